"""N > 1 path on CPU: world_size-2 gloo processes shard the rows, each 'scores' its block with the oracle
(standing in for the per-rank GPU prediction) and the blocks are gathered; the result must equal the
single-process answer.  Also the shard arithmetic (equal and weight-balanced blocks)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    import simspread_jl_amd as ss
    for n in (0, 1, 7, 10, 10_000, 100_003):
        for world in (1, 2, 3, 8):
            blocks = [ss.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_shard_range_weighted_balances_power_law_rows():
    import simspread_jl_amd as ss
    rng = np.random.default_rng(0)
    n, world = 20_000, 8
    w = (np.arange(1, n + 1) ** -1.2)
    w = np.minimum(w / w.mean() * 1000, n)  # Zipf(1.2) row degrees, mean 1000, capped at the column count
    rng.shuffle(w)
    blocks = [ss.shard_range(n, r, world, weights=w) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    loads = np.array([w[a:b].sum() for a, b in blocks])
    assert loads.max() / loads.mean() < 1.25
    eq = np.array([w[a:b].sum() for a, b in (ss.shard_range(n, r, world) for r in range(world))])
    assert loads.max() <= eq.max() + 1e-9


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import simspread_jl_amd as ss
    from oracle import simspread_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Xq, Xs, Ys = O.synth_bipartite(37, 60, 60, 23, 0.2, 0.1, seed=5, dtype=np.float64)
    lo, hi = ss.shard_range(Xq.shape[0], rank, world)
    local = torch.from_numpy(O.predict_factored(Xq[lo:hi], Xs, Ys))   # stand-in for DeviceGraph.predict on this rank
    full = ss.gather_scores(local, Xq.shape[0])
    np.save(os.path.join(tmp, f"full_{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_predict_and_gather(tmp_path):
    import torch.multiprocessing as mp
    from oracle import simspread_oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    Xq, Xs, Ys = O.synth_bipartite(37, 60, 60, 23, 0.2, 0.1, seed=5, dtype=np.float64)
    want = O.predict_factored(Xq, Xs, Ys)
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"full_{r}.npy"), want)
