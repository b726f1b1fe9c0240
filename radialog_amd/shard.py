"""Data-parallel sharding of report generation: every image/report is an independent unit, so a batch is
split into contiguous per-rank shards, each rank runs encode -> prefill -> decode on its own GPU with a full weight
replica and NO data-path collective, and the generated token ids are all-gathered once at the end
(`torch.distributed` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests). The reference has no
inference-time collective at all (its only distributed code is LAVIS' disabled training DDP, runner_base.py:101-118);
this all-gather is the one addition the north star asks for. int32[B_local, N] per rank = 32 KiB at B=32, N=256:
latency-bound on xGMI, one shot."""
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


def allgather_tokens(tokens: torch.Tensor, world: int) -> torch.Tensor:
    """tokens int32[B_local, N] (same shape on every rank) -> int32[world*B_local, N], rank-major."""
    if world == 1:
        return tokens
    import torch.distributed as dist
    tokens = tokens.contiguous()
    out = torch.empty((world * tokens.shape[0],) + tuple(tokens.shape[1:]), dtype=tokens.dtype, device=tokens.device)
    dist.all_gather_into_tensor(out, tokens)
    return out


def allgather_ragged(tokens: torch.Tensor, counts, world: int, pad_id: int = 0) -> torch.Tensor:
    """Shards of unequal size (total not divisible by world): pad each shard to max(counts), gather, drop the padding."""
    if world == 1:
        return tokens
    mx = max(counts)
    if tokens.shape[0] < mx:
        pad = torch.full((mx - tokens.shape[0],) + tuple(tokens.shape[1:]), pad_id, dtype=tokens.dtype, device=tokens.device)
        tokens = torch.cat([tokens, pad], 0)
    g = allgather_tokens(tokens, world).view(world, mx, *tokens.shape[1:])
    return torch.cat([g[r, : counts[r]] for r in range(world)], 0)
