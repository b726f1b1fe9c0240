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


def _worker(rank, world, port, total, n_new, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, world, rank)
        toks = _fake_generate(lo, hi - lo, n_new)
        if total % world == 0:
            out = allgather_tokens(toks, world)
        else:
            counts = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
            out = allgather_ragged(toks, counts, world)
        q.put((rank, out.tolist()))      # by value: a shared-memory tensor would need this process to outlive the read
    finally:
        dist.destroy_process_group()


def _run(total, n_new, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, n_new, q)) for r in range(world)]
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
