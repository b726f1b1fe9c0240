"""pconv_k (round 4: the image encoder's fragment-packed convolution / GEMM family, radialog_amd/csrc/pconv.hip) on its own, through the
C ABI hook rdx_conv_test: every tap / stride / epilogue combination, every register tile incl. the K split inside the workgroup, ragged
last tiles, image borders on every side -- against an fp32 CPU convolution of the same rounded operands (torch.nn.functional.conv2d: the
arithmetic the reference's torchvision Bottleneck / nn.Linear layers run, biovil_t/resnet.py:25-47, Qformer.py:285-375) and against the
row-major kernels of rounds 1-3 on the same data. Rounding points: T(acc + bias) [relu | gelu], then T([relu](resid + that))."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NONE, RELU, GELU, RESID, RESRELU = 0, 1, 2, 3, 6
DT = {"f16": torch.float16, "bf16": torch.bfloat16}
TOL = {"f16": 4e-3, "bf16": 3.2e-2}          # |values| < 4: two model-dtype ulps there (2^-9 / 2^-6 each) -- the residual epilogues round twice


@pytest.fixture(scope="module", params=["bf16", "f16"])
def eng(request):
    from radialog_amd.config import small_cfg
    from radialog_amd.engine import RdxEngine
    e = RdxEngine(small_cfg(), dtype=request.param, device=0, max_batch=1, max_len=32, llama=False, vision=False)
    e.dtype_name = request.param
    yield e
    e.close()


def _ref(x, w, bias, res, k, stride, epi, dt):
    cout, cin = w.shape[0], x.shape[-1]
    wt = w.to(dt).float().view(cout, k, k, cin).permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wt, bias, stride=stride, padding=k // 2).permute(0, 2, 3, 1)
    if epi == RELU:
        y = y.relu()
    elif epi == GELU:
        y = torch.nn.functional.gelu(y)
    elif epi == RESID:
        y = y.to(dt).float() + res.float()
    elif epi == RESRELU:
        y = (y.to(dt).float() + res.float()).relu()
    return y


# (B, H, Cin, Cout, ksize, stride, epilogue): the trunk's shape classes at sizes the CPU reference does in a blink -- 1 x 1, 3 x 3, stride 2 of
# both, the residual tail, odd sizes (61 -> 31: the findings classifier's 488 px crop), few and many K chunks, Cin = 32 (one chunk per tap)
CONV_CASES = [
    (2, 16, 64, 64, 1, 1, RELU), (2, 16, 64, 64, 3, 1, RELU), (2, 16, 64, 256, 1, 1, RESRELU), (2, 16, 64, 256, 1, 1, NONE),
    (1, 28, 128, 128, 3, 2, RELU), (1, 28, 256, 512, 1, 2, NONE), (3, 7, 512, 512, 3, 1, RELU), (1, 14, 1024, 256, 1, 1, RELU),
    (1, 61, 32, 32, 3, 2, RELU), (2, 13, 32, 64, 3, 1, NONE), (1, 31, 64, 64, 3, 1, RELU), (1, 9, 256, 1024, 1, 1, RESRELU),
    # the Q-Former / projector GEMMs as 1 x 1 convolutions (rows = B x H^2): channel counts that are not powers of two, GELU, plain residual
    (2, 4, 768, 2304, 1, 1, NONE), (2, 4, 768, 768, 1, 1, RESID), (2, 4, 768, 3072, 1, 1, GELU), (2, 4, 3072, 768, 1, 1, RESID),
    (1, 14, 1408, 1408, 1, 1, NONE), (4, 4, 352, 96, 1, 1, RELU),
]


@pytest.mark.parametrize("B,H,cin,cout,k,stride,epi", CONV_CASES)
def test_pconv_matches_fp32_convolution_and_the_row_major_kernels(eng, B, H, cin, cout, k, stride, epi, monkeypatch):
    dt = DT[eng.dtype_name]
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + cin + k)
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    K = k * k * cin
    x = (torch.randn(B, H, H, cin, generator=g) * 0.5).to(dt)
    w = torch.randn(cout, K, generator=g) / K ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = (torch.randn(B, Ho, Ho, cout, generator=g) * 0.5).to(dt) if epi in (RESID, RESRELU) else None
    ref = _ref(x, w, bias, res, k, stride, epi, dt)
    tol = TOL[eng.dtype_name] * max(1.0, float(ref.abs().max()) / 4.0)
    monkeypatch.delenv("RDX_PCONV_TILE", raising=False)
    packed = eng.conv_test(x, w, bias, res, k, stride, epi, path=1).float().cpu()          # pack -> pconv_k (own tile choice) -> unpack
    assert torch.isfinite(packed).all()
    e_new = float((packed - ref).abs().max())
    assert e_new < tol, f"pconv_k differs from the fp32 convolution by {e_new} (tolerance {tol})"
    rowout = eng.conv_test(x, w, bias, res, k, stride, epi, path=2).float().cpu()          # pconv_k writing row-major itself
    assert torch.equal(rowout, packed), "row-major-output variant differs from the packed one"
    old = eng.conv_test(x, w, bias, res, k, stride, epi, path=0).float().cpu()             # the row-major kernels of rounds 1-3
    assert float((old - packed).abs().max()) <= tol, "pconv_k and the row-major kernels disagree beyond rounding order"
    # every register tile, with and without the K split inside the workgroup: the same values up to the order of the partial sums
    for tile in ("4x4", "4x2", "2x4", "2x2", "1x2", "4x4k4", "4x2k8", "2x2k8", "1x2k4"):
        if (cout // 16) % int(tile[2]):
            continue
        monkeypatch.setenv("RDX_PCONV_TILE", tile)
        y = eng.conv_test(x, w, bias, res, k, stride, epi, path=1).float().cpu()
        e = float((y - ref).abs().max())
        assert e < tol, f"tile {tile}: differs from the fp32 convolution by {e}"
        assert float((y - packed).abs().max()) <= tol, f"tile {tile}: differs from the default tile beyond rounding order"


def test_pconv_refuses_shapes_it_has_no_kernel_for(eng):
    """Odd column-tile counts (the epilogue pairs 16-channel tiles), channel counts that are not whole 32-deep chunks: rdx_conv_test reports
    it instead of computing something else."""
    from radialog_amd._lib import RdxError
    x = torch.zeros(1, 4, 4, 64, dtype=DT[eng.dtype_name])
    with pytest.raises(RdxError, match="not supported by pconv"):
        eng.conv_test(x, torch.zeros(48, 64), None, None, 1, 1, NONE, path=1)
    x3 = torch.zeros(1, 4, 4, 96, dtype=DT[eng.dtype_name])
    with pytest.raises(RdxError, match="not supported by pconv"):
        eng.conv_test(x3, torch.zeros(64, 9 * 96), None, None, 3, 1, NONE, path=1)       # 3 x 3 needs a power-of-two channel count
