"""How fast the CPU oracle's GEMMs run against torch's intra-op thread count on this host (the GPU boxes have 256 cpus; the default pool
of os.cpu_count() threads is not the fastest): python tools/oracle_threads.py"""
import os, time, torch
import torch.nn.functional as F
print("cpus", os.cpu_count(), "default threads", torch.get_num_threads())
for dt in (torch.float16, torch.bfloat16):
    w = torch.randn(11008, 4096).to(dt)
    for M in (1, 32, 160, 5120):
        x = torch.randn(M, 4096).to(dt)
        row = []
        for thr in (8, 16, 32, 64, 128, 256):
            if thr > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(thr)
            F.linear(x, w)
            n = 3 if M >= 160 else 10
            t0 = time.time()
            for _ in range(n):
                F.linear(x, w)
            row.append("%d:%.1fms" % (thr, (time.time() - t0) / n * 1e3))
        print(str(dt).split(".")[1], "M=%d" % M, "  ".join(row), flush=True)
