"""pconv_k (fragment-packed convolution) against the row-major production kernels: correctness on random data (small batch) and ms per
launch at batch B for every trunk shape. python tools/pconv_check.py [B] [tiles...]"""
import os as _os
_os.environ.setdefault("RDX_DEBUG_HOOKS", "1")      # this tool drives the kernel-test hooks of librdx_hooks.so (include/rdx_hooks.h)
import os, sys, torch
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tiles = sys.argv[2:] or ["auto"]
eng = RdxEngine(small_cfg(), dtype="bf16", device=0, max_batch=1, max_len=32, llama=False, vision=False)
dev = eng.device
RELU, NONE, RESRELU = 1, 0, 6
shapes = [("l1.c1a 64->64", 112, 64, 64, 1, 1, RELU), ("l1.c2 3x3 64", 112, 64, 64, 3, 1, RELU), ("l1.c3 64->256 +res", 112, 64, 256, 1, 1, RESRELU),
          ("l1.ds 64->256", 112, 64, 256, 1, 1, NONE), ("l1.c1b 256->64", 112, 256, 64, 1, 1, RELU),
          ("l2.c1a 256->128", 112, 256, 128, 1, 1, RELU), ("l2.c2a 3x3 s2", 112, 128, 128, 3, 2, RELU), ("l2.ds 256->512 s2", 112, 256, 512, 1, 2, NONE),
          ("l2.c3 128->512 +res", 56, 128, 512, 1, 1, RESRELU), ("l2.c1b 512->128", 56, 512, 128, 1, 1, RELU), ("l2.c2 3x3 128", 56, 128, 128, 3, 1, RELU),
          ("l3.c1a 512->256", 56, 512, 256, 1, 1, RELU), ("l3.c2a 3x3 s2", 56, 256, 256, 3, 2, RELU), ("l3.ds 512->1024 s2", 56, 512, 1024, 1, 2, NONE),
          ("l3.c3 256->1024 +res", 28, 256, 1024, 1, 1, RESRELU), ("l3.c1b 1024->256", 28, 1024, 256, 1, 1, RELU), ("l3.c2 3x3 256", 28, 256, 256, 3, 1, RELU),
          ("l4.c1a 1024->512", 28, 1024, 512, 1, 1, RELU), ("l4.c2a 3x3 s2", 28, 512, 512, 3, 2, RELU), ("l4.ds 1024->2048 s2", 28, 1024, 2048, 1, 2, NONE),
          ("l4.c3 512->2048 +res", 14, 512, 2048, 1, 1, RESRELU), ("l4.c1b 2048->512", 14, 2048, 512, 1, 1, RELU), ("l4.c2 3x3 512", 14, 512, 512, 3, 1, RELU)]
GELU, RESID = 2, 3
# the Q-Former / projector GEMMs as 1 x 1 convolutions over 2 B "images" of 4 x 4 pixels (M = 32 B rows) resp. B images of 14 x 14
gemms = [("q.qkv 768->2304", 4, 768, 2304, NONE, 2), ("q.o 768->768 +res", 4, 768, 768, RESID, 2), ("q.ffn1 768->3072 gelu", 4, 768, 3072, GELU, 2),
         ("q.ffn2 3072->768 +res", 4, 3072, 768, RESID, 2), ("b2v 2048->256", 14, 2048, 256, NONE, 1), ("proj1 256->1408", 14, 256, 1408, RELU, 1),
         ("proj2 1408->1408", 14, 1408, 1408, NONE, 1)]
only = os.environ.get("ONLY", "")
g = torch.Generator(device="cpu").manual_seed(5)
tot = {t: 0.0 for t in ["old"] + tiles}
counts = {"l1.c1a": 1, "l1.c2": 3, "l1.c3": 3, "l1.ds": 1, "l1.c1b": 2, "l2.c1a": 1, "l2.c2a": 1, "l2.ds": 1, "l2.c3": 4, "l2.c1b": 3, "l2.c2": 3,
          "l3.c1a": 1, "l3.c2a": 1, "l3.ds": 1, "l3.c3": 6, "l3.c1b": 5, "l3.c2": 5, "l4.c1a": 1, "l4.c2a": 1, "l4.ds": 1, "l4.c3": 3, "l4.c1b": 2, "l4.c2": 2}
counts.update({"q.qkv": 12, "q.o": 18, "q.ffn1": 12, "q.ffn2": 12, "b2v": 1, "proj1": 1, "proj2": 1})
allshapes = [(n, H, ci, co, k, st, ep, 1) for (n, H, ci, co, k, st, ep) in shapes] + [(n, H, ci, co, 1, 1, ep, mul) for (n, H, ci, co, ep, mul) in gemms]
Bbase = B
for name, H, cin, cout, k, stride, epi, bmul in allshapes:
    if only and only not in name: continue
    B = Bbase * bmul
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    K = k * k * cin
    # correctness at batch 2 (odd sizes exercise the ragged last tile: H -> H - 1 for one leg)
    line = f"{name:24s}"
    for (b, h) in ((2, H), (1, H - 3 if k == 3 and stride == 1 else H)):
        ho = (h + 2 * (k // 2) - k) // stride + 1
        x = (torch.randn(b, h, h, cin, generator=g) * 0.5).to(torch.bfloat16)
        w = torch.randn(cout, K, generator=g) / K ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        res = (torch.randn(b, ho, ho, cout, generator=g) * 0.5).to(torch.bfloat16) if epi in (RESRELU, RESID) else None
        o0 = eng.conv_test(x, w, bias, res, k, stride, epi, 0).float().cpu()
        os.environ.pop("RDX_PCONV_TILE", None)
        o1 = eng.conv_test(x, w, bias, res, k, stride, epi, 1).float().cpu()
        o2 = eng.conv_test(x, w, bias, res, k, stride, epi, 2).float().cpu() if epi in (NONE, RELU, RESRELU) or True else o1
        os.environ["RDX_PCONV_TILE"] = "2x2k4"
        o3 = eng.conv_test(x, w, bias, res, k, stride, epi, 1).float().cpu()                 # the K-split variant on the same data
        os.environ.pop("RDX_PCONV_TILE", None)
        # fp32 reference on the CPU for the small leg only
        ref = None
        if b * h * h * K * cout < 3e10:
            wt = w.to(torch.bfloat16).float().view(cout, k, k, cin).permute(0, 3, 1, 2)
            y = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wt, bias, stride=stride, padding=k // 2).permute(0, 2, 3, 1)
            if epi == RELU: y = y.relu()
            if epi == RESRELU: y = (y.to(torch.bfloat16).float() + res.float()).relu()
            if epi == RESID: y = y.to(torch.bfloat16).float() + res.float()
            if epi == GELU: y = torch.nn.functional.gelu(y)
            ref = y
        d01 = float((o0 - o1).abs().max()); d12 = float((o1 - o2).abs().max())
        dr = float((o1 - ref).abs().max()) if ref is not None else -1
        d0r = float((o0 - ref).abs().max()) if ref is not None else -1
        d13 = float((o1 - o3).abs().max())
        line += f" | b{b} h{h}: |old-new| {d01:.3g} |new-rowout| {d12:.3g} |new-ksplit| {d13:.3g} |new-ref| {dr:.3g} |old-ref| {d0r:.3g}"
    print(line, flush=True)
    # timing at batch B
    x = (torch.randn(B, H, H, cin, generator=g) * 0.5).to(torch.bfloat16)
    w = torch.randn(cout, K, generator=g) / K ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = (torch.randn(B, Ho, Ho, cout, generator=g) * 0.5).to(torch.bfloat16) if epi in (RESRELU, RESID) else None
    fl = 2.0 * B * Ho * Ho * cout * K
    _, ms0 = eng.conv_test(x, w, bias, res, k, stride, epi, 0, iters=10)
    key = name.split()[0]
    tot["old"] += ms0 * counts[key]
    tl = f"    B={B}: old {ms0*1e3:7.1f} us ({fl/ms0/1e9:6.0f} TF)"
    for t in tiles:
        if t == "auto": os.environ.pop("RDX_PCONV_TILE", None)
        else: os.environ["RDX_PCONV_TILE"] = t
        try:
            _, ms1 = eng.conv_test(x, w, bias, res, k, stride, epi, 1, iters=10)
        except Exception as e:
            ms1 = float("nan")
        tot[t] += ms1 * counts[key]
        tl += f" | {t} {ms1*1e3:7.1f} us ({fl/ms1/1e9:6.0f} TF)"
    print(tl, flush=True)
print("trunk totals (ms, launch counts applied):", {k: round(v, 3) for k, v in tot.items()})
eng.close()
