"""Data-parallel sharding of report generation: every image/report is an independent unit, so a batch is
split into contiguous per-rank shards, each rank runs encode -> prefill -> decode on its own GPU with a full weight
replica and NO data-path collective, and the generated token ids are all-gathered once at the end
(one `ncclAllGather` over xGMI issued by librdx itself -- rdx_comm_init / rdx_allgather_tokens of include/rdx.h, on the engine's
own stream; torch.distributed is only the launcher-side plumbing that carries the 128-byte RCCL id from rank 0 to the others, and
the "gloo" backend of the CPU tests). The reference has no inference-time collective at all (its only distributed code is LAVIS'
disabled training DDP, runner_base.py:101-118); this all-gather is the one addition the north star asks for.
int32[B_local, N] per rank = 32 KiB at B=32, N=256: latency-bound on xGMI, one shot."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, end) of `total` items owned by `rank`; earlier ranks take the remainder."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_comm(engine, rank: int, world: int) -> None:
    """Bring up the engine's RCCL communicator: rank 0 draws the unique id (rdx_comm_unique_id), the launcher's process group
    (any backend; only a 128-byte host broadcast) hands it round, every rank calls rdx_comm_init. world = 1 is allowed
    (`torchrun --nproc-per-node 1`): the collective then runs over a single rank."""
    import torch.distributed as dist
    box = [engine.comm_unique_id() if rank == 0 else None]
    if world > 1:
        if not dist.is_initialized():
            raise RuntimeError("init_comm: torch.distributed must be initialised by the launcher to exchange the RCCL id")
        dist.broadcast_object_list(box, src=0)
    engine.comm_init(box[0], rank, world)


def allgather_tokens(tokens: torch.Tensor, world: int, engine=None) -> torch.Tensor:
    """tokens int32[B_local, N] (same shape on every rank) -> int32[world*B_local, N], rank-major. With an `engine` whose
    communicator is up (init_comm) the gather is librdx's ncclAllGather; otherwise torch.distributed (the gloo CPU tests)."""
    if engine is not None and engine.comm_world > 0:
        return engine.allgather_tokens(tokens)
    if world == 1:
        return tokens
    import torch.distributed as dist
    tokens = tokens.contiguous()
    dev = tokens.device
    if tokens.is_cuda and dist.get_backend() == "gloo":     # host process group, device tokens (a launcher whose RCCL bring-up failed): stage through the host
        tokens = tokens.cpu()
    out = torch.empty((world * tokens.shape[0],) + tuple(tokens.shape[1:]), dtype=tokens.dtype, device=tokens.device)
    dist.all_gather_into_tensor(out, tokens)
    return out.to(dev)


def allgather_ragged(tokens: torch.Tensor, counts, world: int, pad_id: int = 0, engine=None) -> torch.Tensor:
    """Shards of unequal size (total not divisible by world): pad each shard to max(counts), gather, drop the padding."""
    if world == 1:
        return tokens
    mx = max(counts)
    if tokens.shape[0] < mx:
        pad = torch.full((mx - tokens.shape[0],) + tuple(tokens.shape[1:]), pad_id, dtype=tokens.dtype, device=tokens.device)
        tokens = torch.cat([tokens, pad], 0)
    g = allgather_tokens(tokens, world, engine).view(world, mx, *tokens.shape[1:])
    return torch.cat([g[r, : counts[r]] for r in range(world)], 0)
