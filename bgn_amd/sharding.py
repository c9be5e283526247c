"""Batch sharding across the GPUs of one node.

Every batch element is independent, so a batch shards by contiguous ranges:
rank g of G takes [g*N/G, (g+1)*N/G).  The key context is replicated (each
process creates its own Engine from the same public key); the only collective
is the gather of result arrays (RCCL all-gather when run under
torch.distributed).  MultPoly shards by polynomial so its segmented GT
reduction stays local to one GPU.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def sharded_apply(op: Callable, total: int, elem_bytes_in: int, elem_bytes_out: int, inputs, world: int, rank: int,
                  dist=None):
    """Run `op(slice_of_each_input) -> uint8 tensor` on this rank's slice and gather
    every rank's result (all_gather; ragged slices are padded to the largest).

    `inputs` are uint8 tensors holding `total` rows of elem_bytes_in bytes.
    Returns a uint8 tensor of total*elem_bytes_out bytes on every rank.
    """
    import torch
    lo, hi = shard_range(total, world, rank)
    mine = op(*[t[lo * elem_bytes_in:hi * elem_bytes_in] for t in inputs])
    if world == 1 or dist is None:
        return mine
    per = -(-total // world)
    pad = torch.zeros(per * elem_bytes_out, dtype=torch.uint8, device=mine.device)
    pad[: mine.numel()] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r, part in enumerate(parts):
        l, h = shard_range(total, world, r)
        out.append(part[: (h - l) * elem_bytes_out])
    return torch.cat(out)
