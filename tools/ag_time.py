import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from radialog_amd import shard
from radialog_amd.config import small_cfg
from radialog_amd.engine import RdxEngine
eng = RdxEngine(small_cfg(), dtype="bf16", device=0, max_batch=2, max_len=64, vision=False)
shard.init_comm(eng, 0, 1)
toks = torch.arange(32 * 256, dtype=torch.int32, device=eng.device).view(32, 256)
for _ in range(3): eng.allgather_tokens(toks)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): out = eng.allgather_tokens(toks)
torch.cuda.synchronize(); print("allgather_tokens (1 rank, int32[32,256]): %.1f us per call" % ((time.perf_counter() - t0) / 50 * 1e6))
eng.close()
