"""bench.py's multi-rank branch end to end on a box without GPUs (VERDICT r4 "next" 6): `python bench.py --gpus 2` with RDX_BENCH_STUB=1
starts its own two ranks (torch.distributed.run, 127.0.0.1, gloo), every rank runs main() with the stub engine of bench.py (sleeps instead of
kernels; the token gather goes through the "communicator is up" branch of shard.allgather_tokens), and rank 0 must print ONE JSON line with
the multi-rank fields the driver's SCALE run reads. The real engine is never involved: this pins control flow, not numbers."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=()):
    env = dict(os.environ)
    env["RDX_BENCH_STUB"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", *extra],
                       env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert p.returncode == 0, f"rc {p.returncode}\nstdout: {p.stdout[-2000:]}\nstderr: {p.stderr[-4000:]}"
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"expected ONE JSON line on stdout, got {len(lines)}: {p.stdout[-1000:]}"
    return json.loads(lines[0]), p.stderr


def test_bench_gpus2_emits_one_line_with_the_multi_rank_fields():
    d, err = _run(2)
    assert "starting 2 ranks" in err                                   # bench.py became its own launcher
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2
    assert len(d["per_rank_ms_per_step"]) == 2
    # the clock is the MAX over ranks (the per-step all-gather keeps the ranks in step, so the entries are close: the stub's rank 1 sleeps
    # twice as long per report and rank 0 waits for it in the collective)
    assert abs(d["ms_per_step"] - max(d["per_rank_ms_per_step"])) < 1e-6 and min(d["per_rank_ms_per_step"]) > 4.0
    assert d["rccl_ranks"] == 2 and "rdx_allgather_tokens" in d["collective"]
    # N > 1: the CPU oracle is not re-timed, the line carries the pointer to the committed N = 1 figure
    assert "source" in d["cpu_baseline"] or d["cpu_baseline"]["value"] is None
    assert d["cpu_baseline"]["kind"] == "port" and "N = 1" in d["cpu_baseline"]["sample"]
    assert d["oracle_checked"] is False
    # both sub-runs ride the same line as scalars (round 6: the line stays under 4 KB; the nested records are in the detail file beside bench.py)
    assert len(json.dumps(d)) < 4096
    assert d["b32_global_batch"] == 64 and d["fp8_b32_global_batch"] == 64 and d["b32_value"] > 0 and d["fp8_b32_value"] > 0
    assert d["roofline"]["b32_decode_avg_frac"] == d["b32_frac"]
    with open(os.path.join(REPO, d["detail"])) as f:
        full = json.load(f)
    b32, f8 = full["b32"], full["fp8_b32"]
    assert b32["workload"].startswith("configs[3]") and "RCCL all-gather" in b32["workload"]
    assert b32["n_gpus"] == 2 and b32["per_gpu_batch"] == 32 and b32["global_batch"] == 64 and len(b32["per_rank_ms_per_step"]) == 2
    assert f8["workload"].startswith("configs[4]") and f8["global_batch"] == 64 and f8["dtype"].endswith("+fp8w")
    assert "value_f16" not in d and "enc_b256" not in d               # N = 1 extras
    assert d["data"].startswith("STUB")                                # a stub line can never be mistaken for a measurement


def test_bench_stub_single_rank_under_the_launcher():
    """The driver's N = 1 form with a launcher in the environment (`torchrun --nproc-per-node 1`): one rank, a 1-rank communicator."""
    env = dict(os.environ)
    env.update(RDX_BENCH_STUB="1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0", "--no-b32", "--no-fp8"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and len(d["per_rank_ms_per_step"]) == 1 and "b32_value" not in d
