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


def _worker(rank, world, port, tmp, use_hip):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import simspread_jl_amd as ss
    from oracle import simspread_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq = 37 if not use_hip else 301                      # uneven blocks in both cases
    Xq, Xs, Ys = O.synth_bipartite(nq, 60 if not use_hip else 900, 60 if not use_hip else 900, 23 if not use_hip else 257,
                                   0.2 if not use_hip else 0.05, 0.1 if not use_hip else 0.03, seed=5,
                                   dtype=np.float64 if not use_hip else np.float32)
    lo, hi = ss.shard_range(Xq.shape[0], rank, world)
    if use_hip:
        # the HIP path on every rank: the ranks share device 0 of a one-GPU box (the 8-GPU run gives each its own);
        # the whole graph is replicated, the rank scores only its block of query rows -- no data-path collective
        ss.init(0)
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
        local = torch.from_numpy(g.predict("query", row_begin=lo, row_end=hi).copy())
        top_i, top_v = ss.topl(local.numpy(), 5)
        top_i, top_v = torch.from_numpy(top_i.copy()), torch.from_numpy(top_v.copy())
    else:
        local = torch.from_numpy(O.predict_factored(Xq[lo:hi], Xs, Ys))   # stand-in for DeviceGraph.predict on this rank
        order = np.argsort(-local.numpy(), axis=1, kind="stable")[:, :5]
        top_i = torch.from_numpy(order.astype(np.int32))
        top_v = torch.from_numpy(np.take_along_axis(local.numpy(), order, axis=1))
    counts = [ss.shard_range(Xq.shape[0], r, world)[1] - ss.shard_range(Xq.shape[0], r, world)[0] for r in range(world)]
    full = ss.gather_scores(local, Xq.shape[0])                                   # every rank gets the matrix
    at_root = ss.gather_scores(local, Xq.shape[0], root=1, counts=counts)           # only rank 1 does
    assert (at_root is None) == (rank != 1)
    if rank == 1:
        assert torch.equal(at_root, full)
    gi, gv = ss.gather_topl(top_i, top_v, Xq.shape[0], counts=counts)              # reduced gather: L numbers per row
    np.save(os.path.join(tmp, f"full_{rank}.npy"), full.numpy())
    np.save(os.path.join(tmp, f"topi_{rank}.npy"), gi.numpy())
    np.save(os.path.join(tmp, f"topv_{rank}.npy"), gv.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_sharded_predict_and_gather(tmp_path):
    import torch.multiprocessing as mp
    from oracle import simspread_oracle as O
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), False), nprocs=2, join=True)
    Xq, Xs, Ys = O.synth_bipartite(37, 60, 60, 23, 0.2, 0.1, seed=5, dtype=np.float64)
    want = O.predict_factored(Xq, Xs, Ys)
    order = np.argsort(-want, axis=1, kind="stable")[:, :5]
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"full_{r}.npy"), want)
        np.testing.assert_array_equal(np.load(tmp_path / f"topi_{r}.npy"), order)
        np.testing.assert_array_equal(np.load(tmp_path / f"topv_{r}.npy"), np.take_along_axis(want, order, axis=1))


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start its two ranks itself (fresh children of
    torch.distributed.run, created before the parent touches a GPU) and hand rank 0's JSON line through; here the ranks
    only rendezvous over gloo on the CPU (--rendezvous-only), which is the whole launch path minus the device work."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["rendezvous"] == {"world": 2, "sum_of_rank_plus_1": 3, "gpus_arg": 2}


def test_bench_self_launch_hands_back_the_exit_code():
    """A failing rank must surface as a non-zero exit code of the plain command (no silent rc 0)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--no-such-flag"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0


def _lib_gather_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import simspread_jl_amd as ss
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # only the 128-byte id travels over this
    ss.init(rank)
    ss.use_torch_stream()
    ids = [ss.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ss.comm_init(ids[0], rank, world)
    counts = [5, 3] if world == 2 else [5 + r for r in range(world)]
    g = torch.Generator(device="cuda")
    g.manual_seed(100 + rank)
    for dtype in (torch.float32, torch.float64):
        local = torch.rand((counts[rank], 257), device="cuda", dtype=dtype, generator=g)
        full = ss.lib_gather_scores(local, counts)
        at_root = ss.lib_gather_scores(local, counts, root=world - 1)
        assert (at_root is None) == (rank != world - 1)
        if at_root is not None:
            assert torch.equal(at_root, full)
        np.save(os.path.join(tmp, f"lib_{dtype}_{rank}.npy".replace("torch.", "")), full.cpu().numpy())
        np.save(os.path.join(tmp, f"loc_{dtype}_{rank}.npy".replace("torch.", "")), local.cpu().numpy())
    ss.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_process_library_rccl_gather_uneven_counts(tmp_path):
    """The in-library gather (ss_gather_rows_*, comm.hip) with two ranks on two devices: uneven row counts, all ranks
    receiving and root-only, fp32 and fp64.  Needs two GPUs (RCCL refuses two ranks on one device): skipped on the
    one-GPU boxes, runs wherever the suite meets a multi-GPU node; bench.py --gpus N runs the same exchange by default."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL does not run two ranks on one device")
    mp.spawn(_lib_gather_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for dt in ("float32", "float64"):
        want = np.concatenate([np.load(tmp_path / f"loc_{dt}_{r}.npy") for r in range(2)])
        for r in range(2):
            assert np.array_equal(np.load(tmp_path / f"lib_{dt}_{r}.npy"), want)


@pytest.mark.gpu
def test_two_rank_hip_path_equals_single_process_bit_for_bit(tmp_path):
    """The N > 1 path with the HIP kernels on every rank (two processes sharing the box's one GPU, gloo for the
    exchange): the gathered matrix must equal the single-process prediction bit for bit -- rows are scored
    independently of how they are blocked -- and the reduced top-L gather must equal top-L of the full matrix."""
    import torch.multiprocessing as mp
    import simspread_jl_amd as ss
    from oracle import simspread_oracle as O
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    Xq, Xs, Ys = O.synth_bipartite(301, 900, 900, 257, 0.05, 0.03, seed=5, dtype=np.float32)
    ss.init(0)
    single = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32).predict("query").copy()
    ti, tv = ss.topl(single, 5)
    want = O.predict_factored(Xq.astype(np.float64), Xs.astype(np.float64), Ys.astype(np.float64))
    assert np.abs(single - want).max() / np.abs(want).max() < 1e-5
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"full_{r}.npy"), single)
        assert np.array_equal(np.load(tmp_path / f"topi_{r}.npy"), ti)
        assert np.array_equal(np.load(tmp_path / f"topv_{r}.npy"), tv)
