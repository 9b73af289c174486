"""Fold / query-row sharding across ranks (one process per GPU) and the final score gather.

Rows of the score matrix are independent given the replicated graph, so the
data path needs no collective: rank r scores the rows of ``shard_range``.  The only exchange is the
optional gather of the finished score blocks (RCCL over xGMI when the tensors are on GPUs; gloo on CPU
in the tests)."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int, weights: Optional[Sequence[float]] = None) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of 0..n for `rank`.  Without weights: equal row counts (first n % world
    ranks get one more).  With per-row work weights (e.g. nnz of the row, for power-law graphs): blocks of
    equal total weight (prefix-sum split)."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    if weights is None:
        base, rem = divmod(n, world)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)
    w = np.asarray(weights, dtype=np.float64)
    if w.shape != (n,):
        raise ValueError("weights must have one entry per row")
    c = np.concatenate([[0.0], np.cumsum(w)])
    total = c[-1]
    cuts = [int(np.searchsorted(c, total * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n
    for i in range(1, world + 1):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def gather_scores(local, n_total: int, group=None):
    """All-gather row blocks of the score matrix (torch tensors, equal column count, possibly unequal row
    counts) into the full (n_total x nt) matrix on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    counts = [int(c.item()) for c in counts]
    if sum(counts) != n_total:
        raise ValueError("row blocks do not add up to the full matrix")
    mx = max(counts)
    pad = torch.zeros((mx, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
