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


def _row_counts(nrows: int, device, group=None):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([nrows], dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    return [int(c) for c in counts.tolist()]


def gather_scores(local, n_total: int, group=None, root: Optional[int] = None, counts: Optional[Sequence[int]] = None):
    """Final score gather: row blocks of the score matrix (torch tensors, same column count, row counts given by the
    sharding, possibly unequal) -> the full (n_total x nt) matrix.

    A DIRECT exchange of exact row counts, no padding and no ring: every rank posts one receive per peer straight
    into its slice of the result and one send of its own block per peer (grouped point-to-point operations;
    ncclSend/ncclRecv on RCCL).  On MI355X the GPUs of a node are fully connected by point-to-point xGMI links
    (7 x ~153 GB/s per GPU), so the exchange drives all links at once, where a ring all-gather is bound by one link
    (~33 ms against ~229 ms for the 40 GB of BASELINE configs[2]).

    root=None: every rank gets the full matrix (all-gather).  root=r: only rank r receives (others return None).
    counts: row count per rank when the caller knows the sharding (saves one tiny collective)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if counts is None:
        counts = _row_counts(local.shape[0], local.device, group)
    counts = [int(c) for c in counts]
    if len(counts) != world or counts[rank] != local.shape[0] or sum(counts) != n_total:
        raise ValueError("row blocks do not add up to the full matrix")
    starts = np.concatenate([[0], np.cumsum(counts)])
    local = local.contiguous()
    receives = root is None or rank == root
    full = torch.empty((n_total, local.shape[1]), dtype=local.dtype, device=local.device) if receives else None
    ops = []
    if receives:
        full[starts[rank]:starts[rank + 1]] = local
        for peer in range(world):
            if peer != rank and counts[peer] > 0:
                ops.append(dist.P2POp(dist.irecv, full[starts[peer]:starts[peer + 1]], _global_rank(peer, group), group))
    if counts[rank] > 0:
        for peer in (range(world) if root is None else [root]):
            if peer != rank:
                ops.append(dist.P2POp(dist.isend, local, _global_rank(peer, group), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full


def _global_rank(group_rank: int, group=None) -> int:
    import torch.distributed as dist
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def gather_topl(idx, val, n_total: int, group=None, root: Optional[int] = None, counts: Optional[Sequence[int]] = None):
    """Reduced gather for ranked evaluation: instead of the rows x targets score block every rank contributes the
    top-L (column, score) pairs of its rows (DeviceGraph / ss_topl_f32), L numbers per row instead of nt -- at
    BASELINE configs[2] 80 MB instead of 40 GB for L = 100.  Same direct exchange as gather_scores."""
    return (gather_scores(idx, n_total, group=group, root=root, counts=counts),
            gather_scores(val, n_total, group=group, root=root, counts=counts))


# ---------------------------------------------------------------------------------------------- in-library gather
def comm_unique_id() -> bytes:
    """128 bytes created by rank 0 (ss_comm_unique_id); hand them to the other ranks over any channel."""
    import ctypes as C
    from . import _lib as L
    buf = C.create_string_buffer(128)
    L.check(L.lib().ss_comm_unique_id(C.cast(buf, C.c_void_p)))
    return buf.raw


def comm_init(unique_id: bytes, rank: int, nranks: int) -> None:
    """Create this process's RCCL communicator inside the library (after ss.init(device))."""
    import ctypes as C
    from . import _lib as L
    if len(unique_id) != 128:
        raise ValueError("the unique id is 128 bytes")
    buf = C.create_string_buffer(unique_id, 128)
    L.check(L.lib().ss_comm_init(C.cast(buf, C.c_void_p), int(rank), int(nranks)))


def comm_destroy() -> None:
    from . import _lib as L
    L.check(L.lib().ss_comm_destroy())


def lib_gather_scores(local, counts: Sequence[int], root: Optional[int] = None):
    """gather_scores through the library's own RCCL communicator (ss_gather_rows_*): torch CUDA tensors in, the full
    matrix (or None on non-receiving ranks) out.  Same direct exchange, no torch.distributed involved."""
    import ctypes as C
    import torch
    from . import _lib as L
    if not (local.is_cuda and local.is_contiguous()):
        raise TypeError("local must be a contiguous CUDA tensor")
    # `local` was produced on torch's stream and `full` is allocated by torch: the exchange must be enqueued on that same
    # stream, or nothing orders it behind the kernels that wrote `local`
    L.use_torch_stream()
    rank, nranks = C.c_int(), C.c_int()
    L.check(L.lib().ss_comm_info(C.byref(rank), C.byref(nranks)))
    if nranks.value > 0 and (nranks.value != len(counts) or counts[rank.value] != local.shape[0]):
        raise ValueError("counts do not match the communicator / the local block")
    receives = root is None or root == rank.value
    full = torch.empty((int(sum(counts)), local.shape[1]), dtype=local.dtype, device=local.device) if receives else None
    cnt = (C.c_int64 * len(counts))(*[int(c) for c in counts])
    fn = L.lib().ss_gather_rows_f32 if local.dtype == torch.float32 else L.lib().ss_gather_rows_f64
    L.check(fn(local.data_ptr(), local.shape[1], cnt, None if full is None else full.data_ptr(), -1 if root is None else int(root)))
    L.check(L.lib().ss_synchronize())
    return full
