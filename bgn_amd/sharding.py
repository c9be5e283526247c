"""Batch sharding across the GPUs of one node, multi-process form.

Every batch element is independent, so a batch shards by contiguous ranges:
rank g of G takes [g*N/G, (g+1)*N/G) — the split of bgn_shard_range in the C
ABI, which the one-process form (bgn_mctx_*, MultiEngine) uses too.  The key
context is replicated (each process creates its own Engine from the same public
key); the only collective is the gather of result arrays (RCCL all-gather under
torch.distributed's "nccl" backend, gloo in the CPU tests).  MultPoly shards by
polynomial so its d1*d2 pairings and their GT accumulation stay on one GPU
(poly.go:139-153).

`ShardedOps` is what bench.py and the tests drive: it takes any object with the
Engine's host-buffer methods (mult / poly_mult / decrypt / add) and runs this
rank's shard of a global batch through it.
"""
from __future__ import annotations

from typing import Callable, Sequence, Tuple


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units owned by `rank`; sizes differ by at most one
    (same arithmetic as bgn_shard_range, csrc/multi.cpp)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_shards(mine, total: int, unit_bytes: int, world: int, rank: int, dist=None):
    """All-gather of the per-rank result arrays (uint8 tensors of (hi-lo)*unit_bytes bytes) into the full array
    of total*unit_bytes bytes on every rank.  Equal shards go through all_gather_into_tensor (one RCCL
    all-gather); ragged ones are padded to the largest shard."""
    import torch
    if world == 1 or dist is None:
        return mine
    if total % world == 0:
        out = torch.empty(total * unit_bytes, dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(out, mine.contiguous())
        return out
    per = -(-total // world)
    pad = torch.zeros(per * unit_bytes, dtype=torch.uint8, device=mine.device)
    pad[: mine.numel()] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r, part in enumerate(parts):
        lo, hi = shard_range(total, world, r)
        out.append(part[: (hi - lo) * unit_bytes])
    return torch.cat(out)


def sharded_apply(op: Callable, total: int, unit_bytes_in, unit_bytes_out: int, inputs: Sequence, world: int,
                  rank: int, dist=None):
    """Run `op(slice_of_each_input) -> uint8 tensor` on this rank's slice of `total` units and gather every rank's
    result.  `inputs` are uint8 tensors holding `total` units of unit_bytes_in bytes (one int, or one per input).
    Returns a uint8 tensor of total*unit_bytes_out bytes on every rank."""
    lo, hi = shard_range(total, world, rank)
    ub = [unit_bytes_in] * len(inputs) if isinstance(unit_bytes_in, int) else list(unit_bytes_in)
    mine = op(*[t[lo * u:hi * u] for t, u in zip(inputs, ub)])
    return gather_shards(mine, total, unit_bytes_out, world, rank, dist)


class ShardedOps:
    """This rank's view of a global batch: every method takes the FULL arrays (uint8 tensors on the engine's
    side of the boundary: CPU tensors for the host-buffer entry points), computes the rank's shard with
    `engine` and returns the gathered full result on every rank.

    engine: bgn_amd.Engine — or any object with the same mult / add / poly_mult / decrypt methods over
    byte buffers (the CPU tests pass a stand-in behind that interface)."""

    def __init__(self, engine, elem_bytes: int, world: int, rank: int, dist=None):
        self.engine, self.E, self.world, self.rank, self.dist = engine, int(elem_bytes), int(world), int(rank), dist

    @staticmethod
    def _t(buf):
        import numpy as np
        import torch
        arr = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
        return torch.from_numpy(np.ascontiguousarray(arr).reshape(-1).copy())

    def mult(self, a, b):
        """pk.Mult over `count` pairs (bgn.go:294-314), sharded by element."""
        total = a.numel() // self.E
        return sharded_apply(lambda x, y: self._t(self.engine.mult(x.numpy(), y.numpy())), total, self.E, self.E,
                             [a, b], self.world, self.rank, self.dist)

    def add(self, level: int, a, b):
        total = a.numel() // self.E
        return sharded_apply(lambda x, y: self._t(self.engine.add(level, x.numpy(), y.numpy())), total, self.E, self.E,
                             [a, b], self.world, self.rank, self.dist)

    def poly_mult(self, npoly: int, d1: int, d2: int, a, b):
        """pk.MultPoly over npoly pairs of coefficient vectors (poly.go:123-156), sharded by polynomial."""
        E = self.E

        def op(x, y):
            n_here = x.numel() // (d1 * E)
            return self._t(self.engine.poly_mult(n_here, d1, d2, x.numpy(), y.numpy()))

        return sharded_apply(op, npoly, [d1 * E, d2 * E], (d1 + d2) * E, [a, b], self.world, self.rank, self.dist)

    def decrypt(self, level: int, ct):
        """sk.Decrypt over `count` ciphertexts (bgn.go:205-250): returns (m as int64, status as uint8)."""
        import numpy as np
        import torch
        total = ct.numel() // self.E

        def op(x):
            m, st = self.engine.decrypt(level, x.numpy())
            packed = torch.empty(len(m), 9, dtype=torch.uint8)
            if len(m):                       # (a rank of a batch smaller than the world owns nothing)
                packed[:, :8] = torch.from_numpy(np.ascontiguousarray(m, dtype=np.int64)).view(torch.uint8).reshape(-1, 8)
                packed[:, 8] = torch.from_numpy(np.ascontiguousarray(st, dtype=np.uint8))
            return packed.reshape(-1)

        full = sharded_apply(op, total, self.E, 9, [ct], self.world, self.rank, self.dist).reshape(total, 9)
        return full[:, :8].contiguous().view(torch.int64).reshape(-1), full[:, 8].contiguous()
