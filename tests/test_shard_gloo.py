"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, 'generate' locally (a deterministic stand-in for the
GPU engine -- the collective and the sharding are what is under test) and all-gather the token ids."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radialog_amd.shard import allgather_ragged, allgather_tokens, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_generate(first, count, n_new):
    base = torch.arange(first, first + count, dtype=torch.int32)[:, None]
    return base * 1000 + torch.arange(n_new, dtype=torch.int32)[None]


class _EngineStub:
    """What shard.allgather_tokens / allgather_ragged need of an RdxEngine whose communicator is up: `comm_world` and
    `allgather_tokens` (librdx's ncclAllGather on the GPU; here the same rank-major contract over the gloo group), so that the
    `engine=` branch of the sharding code runs on CPU with two real ranks."""

    def __init__(self, world):
        self.comm_world = world
        self.calls = 0

    def allgather_tokens(self, tokens):
        assert tokens.dtype == torch.int32 and tokens.dim() == 2
        self.calls += 1
        out = torch.empty(self.comm_world * tokens.shape[0], tokens.shape[1], dtype=torch.int32)
        dist.all_gather_into_tensor(out, tokens.contiguous())
        return out


def _worker(rank, world, port, total, n_new, q, use_engine=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, world, rank)
        toks = _fake_generate(lo, hi - lo, n_new)
        eng = _EngineStub(world) if use_engine else None
        if total % world == 0:
            out = allgather_tokens(toks, world, engine=eng)
        else:
            counts = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
            out = allgather_ragged(toks, counts, world, engine=eng)
        if use_engine:
            assert eng.calls == 1, "the engine's collective was bypassed"
        q.put((rank, out.tolist()))      # by value: a shared-memory tensor would need this process to outlive the read
    finally:
        dist.destroy_process_group()


def _run(total, n_new, world=2, use_engine=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, n_new, q, use_engine)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_allgather_tokens_world2():
    res = _run(total=8, n_new=16)
    expect = _fake_generate(0, 8, 16).tolist()
    assert res[0] == expect and res[1] == expect


def test_allgather_ragged_world2():
    res = _run(total=7, n_new=5)
    expect = _fake_generate(0, 7, 5).tolist()
    assert res[0] == expect and res[1] == expect


def test_allgather_through_the_engine_branch_world2():
    """shard.allgather_tokens(engine=...) with a communicator of 2 ranks: the call must go through the engine's collective (on the
    GPU: rdx_allgather_tokens) and keep the rank-major layout; equal and ragged shards."""
    expect = _fake_generate(0, 8, 16).tolist()
    res = _run(total=8, n_new=16, use_engine=True)
    assert res[0] == expect and res[1] == expect
    expect = _fake_generate(0, 7, 5).tolist()
    res = _run(total=7, n_new=5, use_engine=True)
    assert res[0] == expect and res[1] == expect


def _clock_worker(rank, world, port, q):
    import importlib.util
    import sys
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(repo, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        argv, sys.argv = sys.argv, ["bench.py"]
        try:
            spec.loader.exec_module(mod)
        finally:
            sys.argv = argv
        q.put((rank, mod.gather_clock(dist, 1.0 + rank, world, "cpu")))
    finally:
        dist.destroy_process_group()


def test_bench_clock_is_the_max_over_ranks_and_lists_every_rank():
    """bench.py's timing contract at N > 1 (world 4 here): `elapsed` = MAX of the per-rank clocks, `per_rank_ms_per_step` has one entry
    per rank in rank order -- on every rank."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_clock_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        mx, per = res[r]
        assert mx == 4.0 and per == [1.0, 2.0, 3.0, 4.0]
