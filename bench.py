#!/usr/bin/env python3
"""Benchmark of the SimSpread hot path on MI355X.

One "step" = one full predict() pass (stage 1 transfer block + stage 2 W*R SpMM) over one batch of
synthetic input that is already resident in HBM: BASELINE.json configs[1]
    10k queries x 10k targets, 10k sources/features, 5 % similarity, 1 % bipartite density, fp32.
Multi-GPU (torch.distributed, one rank per GPU): query rows are independent, so every rank scores its
own 10k-query shard against the replicated source/target graph (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0 (driver contract).  Besides the headline it carries
    roofline        the time-dominant kernel of the step (stage 1, transfer_kernel) against the HBM roofline on
                    algorithmic bytes, with what actually limits it
    roofline_spmm   the W*R SpMM of the step (stage 2, B = 10^4 columns): fp32 FMA rate and HBM fraction
    spmm_narrow_sweep  the W*R SpMM at the north-star size (100k x 100k, 1 %) for B = 1 .. 64: HBM fractions
    c3_loo          BASELINE configs[2] (100k x 100k, 1 %, leave-one-out) fold throughput of THIS run's ranks,
                    without and with the final score gather -- the curve the north star asks for
    cpu_baseline    the reference algorithm on the host cores: the factored C/OpenMP port on the full workload
                    and the literal dense A*(W*W) (numpy / OpenBLAS dgemm, what SimSpread.jl executes) down-scaled
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0 # ibid.: what streaming kernels reach (SURVEY.md 8d asks for the fraction of this too)
FP32_PEAK_TFLOPS = 157.3    # fp32 vector == fp32-input MFMA peak


def synth_c2(nq, ns, nf, nt, dx, dy, seed, rank, weighted=True):
    """Seeded C2-shaped inputs.  The graph (Xs, Ys) is the same on every rank, the
    query block differs per rank."""
    import scipy.sparse as sp
    rng_g = np.random.default_rng(seed)
    rng_q = np.random.default_rng(seed + 1000 + rank)

    def rand_csr(rng, r, c, d, weighted):
        m = sp.random(r, c, density=d, format="csr", random_state=rng, dtype=np.float32)
        m.data = (0.5 + 0.5 * (1.0 - rng.random(m.nnz))).astype(np.float32) if weighted else np.ones(m.nnz, np.float32)
        m.sort_indices()
        return m

    Xs = rand_csr(rng_g, ns, nf, dx, weighted).tolil()
    Xs.setdiag(1.0)
    Xs = Xs.tocsr().astype(np.float32)
    Xs.sort_indices()
    Ys = rand_csr(rng_g, ns, nt, dy, False)
    Xq = rand_csr(rng_q, nq, nf, dx, weighted)
    return Xq, Xs, Ys


def pmc_entry(kernel_substr):
    """Counters of a kernel from the committed rocprofv3 PMC passes (profiles/pmc_latest.json: FETCH_SIZE and
    WRITE_SIZE collected in separate passes, tools/profile.sh).  bench.py cannot run the profiler on itself, so
    these are the figures of the profiled run of the same command; they are only handed on when the kernel
    sources are the ones that were profiled (source hash recorded with the profile), otherwise null."""
    try:
        from simspread_jl_amd import _lib
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        if d.get("source_sha") != _lib.source_hash():
            return None, "profiles/pmc_latest.json was taken from other kernel sources (sha %s, running %s): dropped" % (
                d.get("source_sha"), _lib.source_hash())
        for k, v in d["kernels"].items():
            if kernel_substr in k:
                return v, "profiles/pmc_latest.json@%s" % d.get("source_sha")
    except Exception as e:  # missing / unreadable profile: no traffic figure
        return None, "no profile (%s)" % type(e).__name__
    return None, "kernel not in profiles/pmc_latest.json"


def csr_bytes(nnz, rows, vb=4):
    return nnz * (vb + 4) + (rows + 1) * 4


def spmm_sweep(ss, torch, steps=5):
    """Narrow-R regime of the W*R SpMM at the north-star size: W 100k x 100k, 1 % (nnz ~1e8), fp32,
    B in {1,2,4,8,16,32,64}; HBM roofline fraction on algorithmic bytes (CSR at 8 B/nnz) and on the
    bytes the kernels really stream (chunk-major operand: 2-byte local index + 4-byte value = 6 B/nnz)."""
    import ctypes as C
    from simspread_jl_amd import _lib as L
    M = K = 100_000
    dens = 0.01
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(20250222 + 3)
    n_draw = int(M * K * dens)
    rows = torch.randint(0, M, (n_draw,), device=dev, generator=g, dtype=torch.int64)
    cols = torch.randint(0, K, (n_draw,), device=dev, generator=g, dtype=torch.int64)
    keys = torch.unique(rows * K + cols)  # sorted, duplicates removed
    del rows, cols
    r = torch.div(keys, K, rounding_mode="floor")
    idx = (keys - r * K).to(torch.int32)
    ptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(r, minlength=M), 0)
    nnz = int(keys.numel())
    del keys, r
    val = torch.rand(nnz, device=dev, dtype=torch.float32, generator=g) + 0.5
    if os.environ.get("SWEEP_BINARY") == "1":   # pattern-only operand (SimSpread's own W = Ys' is binary)
        val.fill_(1.0)
    lib = L.lib()
    h = C.c_void_p()
    L.check(lib.ss_spmat_create_csr_f32(M, K, ptr.data_ptr(), idx.data_ptr(), val.data_ptr(), 0, L.SS_MEM_DEVICE, C.byref(h)))
    out = []
    widths = tuple(int(x) for x in os.environ.get("SWEEP_B", "1,2,4,8,16,32,64").split(","))
    for B in widths:
        R = torch.rand(K, B, device=dev, dtype=torch.float32, generator=g)
        F = torch.empty(M, B, device=dev, dtype=torch.float32)
        ms = []
        for it in range(steps + 2):
            L.check(lib.ss_spmm_f32(h, R.data_ptr(), B, B, 0, F.data_ptr(), B, 0, L.SS_MEM_DEVICE))
            if it >= 2:
                tl = ss.timing_last()
                ms.append(tl["spmm_ms"] + tl["epilogue_ms"])  # layout transposes of the wide path count too
        t = float(np.mean(ms)) * 1e-3
        by = csr_bytes(nnz, M) + K * B * 4 + M * B * 4
        streamed = nnz * 6 + K * B * 4 + M * B * 4
        out.append({"B": B, "ms": round(t * 1e3, 4), "GBps": round(by / t / 1e9, 1),
                    "frac_hbm": round(by / t / 1e9 / HBM_PEAK_GBS, 4), "bytes": by,
                    "frac_hbm_streamed": round(streamed / t / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_hbm_achievable": round(by / t / 1e9 / HBM_ACHIEVABLE_GBS, 4), "kernel": ",".join(ss.path_last())})
    lib.ss_spmat_destroy(h)
    return {"workload": f"W 100k x 100k, 1 percent dense (nnz {nnz}), fp32, CSR streamed once from HBM",
            "frac_hbm": "algorithmic bytes (CSR 8 B/nnz + R + F) / time / 8 TB/s",
            "frac_hbm_streamed": "bytes of the operand the kernels read (6 B/nnz: 2-byte local index + 4-byte value; the compact sliced-ELL operand of B >= 5 adds 0.2-0.6 B/nnz of pair padding and descriptors) + R + F) / time / 8 TB/s",
            "frac_hbm_achievable": "algorithmic bytes / time / 6.3 TB/s (the achievable rate MI355X_MICROARCH.md quotes; SURVEY.md 8d)",
            "results": out}


def build_c3(ss, torch):
    from tools.c3_loo import rand_csr, rand_sym_csr
    n = 100_000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250222 + 3)   # every rank builds the same (replicated) graph
    xp, xi = rand_sym_csr(n, 0.01, gen)
    yp, yi = rand_csr(n, n, 0.01, gen)
    xv = (0.5 + 0.5 * torch.rand(xi.numel(), device="cuda", generator=gen)).float()
    g = ss.DeviceGraph.from_device_csr(0, n, n, n, None, (xp, xi, xv), (yp, yi, None), dtype=np.float32)
    return g, n


def make_agree(torch, dist, world, backend):
    """agree(ok) -> True only if every rank says ok (one tiny MIN all-reduce).  Called after the rank-local part of a leg
    and before its exchange, so that a rank that failed locally does not let its peers walk into a collective alone."""
    def agree(ok=True):
        if world == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))
    return agree


def lib_comm_start(ss, dist, world, rank):
    """The library's own RCCL communicator (ss_comm_*; what a Julia caller uses): rank 0's unique id travels over
    torch.distributed's store, every rank joins."""
    ids = [ss.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ss.comm_init(ids[0], rank, world)


def lib_gather_check(ss, torch, out, world, rank):
    """Before anything is timed: the in-library gather (ss_gather_rows_f32, grouped ncclSend/ncclRecv on the library's
    communicator) against the torch.distributed exchange, bit for bit, on a small block with UNEVEN row counts -- all
    ranks receiving, and rank (world-1) only."""
    counts = [3 + 2 * r for r in range(world)]
    blk = out[:counts[rank]].contiguous()
    tot = sum(counts)
    a = ss.gather_scores(blk, tot, counts=counts)
    b = ss.lib_gather_scores(blk, counts)
    same_all = bool(torch.equal(a, b))
    root = world - 1
    ar = ss.gather_scores(blk, tot, root=root, counts=counts)
    br = ss.lib_gather_scores(blk, counts, root=root)
    same_root = (ar is None and br is None) if rank != root else bool(torch.equal(ar, br))
    return same_all and same_root, counts


def c3_loo_curve(args, ss, torch, dist, world, rank, backend, agree=None, state=None):
    """BASELINE configs[2] on this run's ranks: 100k x 100k, 1 %, leave-one-out; the 10^5 folds are block-sharded
    (ss.shard_range), every rank scores `--folds` consecutive folds of its shard per step (same per-rank work at every
    N: weak scaling in the step, i.e. the full 10^5-fold job gets N times faster).  Reported without any exchange, with
    the direct all-to-all gather of the score blocks (torch.distributed AND, on RCCL, the library's own communicator),
    with a gather to rank 0 only, and with the reduced gather of the top-L predictions per fold."""
    agree = agree or make_agree(torch, dist, world, backend)
    state = state if state is not None else {}
    steps = args.c3_steps
    try:
        g, n = build_c3(ss, torch)
        lo, hi = ss.shard_range(n, rank, world)
        folds = min(args.folds, hi - lo)
        out = torch.empty((folds, n), dtype=torch.float32, device="cuda")
        g.predict_loo(lo, lo + folds, clean=True, out=out)   # warm (operands are cut at first use)
        torch.cuda.synchronize()
        ok = True
    except Exception as e:   # rank-local set-up failed: say so everywhere before anybody enters an exchange
        ok, err = False, repr(e)
    if not agree(ok):
        return {"error": "set-up failed on a rank (%s)" % (err if not ok else "another rank")}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(name, exchange):
        """One timed leg; returns (seconds, stage timings) or None when a rank failed (agreed on by all ranks)."""
        state["leg"] = "c3_loo:" + name
        el = t = None
        try:
            if exchange:
                exchange()
            barrier()
            ss.timing_hold(True)
            t0 = time.perf_counter()
            for i in range(steps):
                b = lo + (i * folds) % max(1, (hi - lo) - folds + 1)
                g.predict_loo(b, b + folds, clean=True, out=out)
                if exchange:
                    exchange()
            barrier()
            el = time.perf_counter() - t0
            t = ss.timing_last()
            ss.timing_hold(False)
            good = True
        except Exception as e:
            good = False
            state.setdefault("errors", []).append("%s on rank %d: %r" % (name, rank, e))
        if not agree(good):
            return None
        if world > 1:
            te = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        return el, t

    counts = [folds] * world

    def ex_all():
        src = out if backend == "nccl" else out.cpu()
        ss.gather_scores(src, folds * world, counts=counts)

    def ex_root():
        src = out if backend == "nccl" else out.cpu()
        ss.gather_scores(src, folds * world, root=0, counts=counts)

    def ex_topl():
        ti, tv = ss.topl(out, 100)
        if backend != "nccl":
            ti, tv = ti.cpu(), tv.cpu()
        ss.gather_topl(ti, tv, folds * world, counts=counts)

    r0 = run("no_exchange", None)
    if r0 is None:
        return {"error": "the leg without exchange failed", "errors": state.get("errors")}
    el0, t = r0
    res = {"workload": "BASELINE configs[2]: 100k x 100k, 1%% density, leave-one-out, folds block-sharded over ranks; %d folds "
                       "per rank and step, %d steps" % (folds, steps),
           "nnz_X": g.nnz_xs, "nnz_Y": g.nnz_ys, "folds_per_rank_step": folds,
           "folds_per_s": folds * world * steps / el0, "edges_per_s": folds * world * steps * n / el0,
           "ms_per_step": el0 / steps * 1e3,
           "stage1_ms": t["transfer_ms"] / max(1, t["transfer_launches"]), "stage2_ms": t["spmm_ms"] / max(1, t["spmm_launches"]),
           "full_loo_seconds_at_this_rate": n / (folds * world * steps / el0)}

    def put(key, r, extra=None):
        if r is None:
            res[key] = {"error": "a rank failed in this leg", "errors": state.get("errors")}
            return False
        res[key] = {"folds_per_s": folds * world * steps / r[0], "ms_per_step": r[0] / steps * 1e3}
        res[key].update(extra or {})
        return True

    alive = True
    if world > 1:
        alive = put("with_gather_all_ranks", run("gather_all_ranks", ex_all),
                    {"bytes_received_per_rank_step": folds * (world - 1) * n * 4,
                     "how": "direct point-to-point exchange of exact row blocks (torch.distributed batch_isend_irecv: grouped "
                            "ncclSend/ncclRecv on RCCL)"})
        alive = alive and put("with_gather_to_rank0", run("gather_to_rank0", ex_root))
    if alive and world > 1 and backend == "nccl" and os.environ.get("BENCH_LIB_GATHER", "1") != "0":
        # the same exchange through the library's own RCCL communicator (ss_comm_init / ss_gather_rows_f32) -- what a
        # Julia caller uses; on by default, checked bit for bit against the torch exchange before it is timed
        state["leg"] = "c3_loo:library_rccl_gather_check"
        try:
            lib_comm_start(ss, dist, world, rank)
            same, small = lib_gather_check(ss, torch, out, world, rank)
            good = True
        except Exception as e:
            good, same, small = False, False, None
            state.setdefault("errors", []).append("library gather on rank %d: %r" % (rank, e))
        if agree(good):
            same = agree(same)
            ok4 = put("with_gather_all_ranks_in_library_rccl", run("gather_all_ranks_library", lambda: ss.lib_gather_scores(out, counts)),
                      {"bitwise_equal_to_torch_exchange": same, "checked_on_row_counts": small,
                       "how": "ss_gather_rows_f32: one ncclRecv per peer into its slice + one ncclSend per peer inside one "
                              "ncclGroupStart/End on the library's communicator (comm.hip)"})
            alive = alive and ok4
            try:
                ss.comm_destroy()
            except Exception:
                pass
        else:
            res["with_gather_all_ranks_in_library_rccl"] = {"error": "communicator set-up or check failed", "errors": state.get("errors")}
            alive = False
    if alive:
        put("with_topL_reduction_L100", run("topL", ex_topl if world > 1 else (lambda: ss.topl(out, 100))),
            {"bytes_received_per_rank_step": folds * (world - 1) * 100 * 8,
             "how": "ss_topl_f32 on the device, then the same exchange on 100 (column, score) pairs per fold"})
    g.close()
    return res


def build_c5(ss, torch, n=100_000):
    """BASELINE configs[4] at its specified weight: 10^5 sources + 10^5 targets, source degrees and target
    popularity Zipf(1.2) with mean 1000 after de-duplication (tools/c5_powerlaw.py), X as C3.  Same bytes on every rank."""
    from tools.c3_loo import rand_sym_csr
    from tools.c5_powerlaw import zipf_bipartite_spec
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250222 + 5)
    xp, xi = rand_sym_csr(n, 0.01, gen)
    yp, yi = zipf_bipartite_spec(n, n, 1000.0, 1.2, gen)
    xv = (0.5 + 0.5 * torch.rand(xi.numel(), device="cuda", generator=gen)).float()
    g = ss.DeviceGraph.from_device_csr(0, n, n, n, None, (xp, xi, xv), (yp, yi, None), dtype=np.float32)
    # work of fold i: stage 1 walks the feature columns of row i of X (sum of their lengths); stage 2 is the same for
    # every fold (one column of R against all of W = Y')
    collen = torch.bincount(xi.long(), minlength=n).double()
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (xp[1:] - xp[:-1]))
    w1 = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, rows, collen[xi.long()])
    ymax_row = int((yp[1:] - yp[:-1]).max().item())
    ymax_col = int(torch.bincount(yi.long(), minlength=n).max().item())
    return g, n, w1.cpu().numpy(), int(yi.numel()), ymax_row, ymax_col


def c5_loo_leg(args, ss, torch, dist, world, rank, backend, agree=None, state=None):
    """BASELINE configs[4] (power-law, 200k nodes): the leave-one-out folds of the whole graph are sharded over this run's
    ranks by ss.shard_range(weights=per-fold work) -- the nnz-balanced partition of SURVEY.md 8(e) -- and, for
    comparison, by equal fold counts.  Every rank scores ITS WHOLE SHARD (strong scaling: the job is the full LOO, capped
    by --c5-folds), `--folds` folds per launch; wall = slowest rank.  No exchange: scores are reduced where they are made."""
    agree = agree or make_agree(torch, dist, world, backend)
    state = state if state is not None else {}
    state["leg"] = "c5_loo:set-up"
    try:
        g, n, w1, nnz_y, ymax_row, ymax_col = build_c5(ss, torch)
        total = min(n, args.c5_folds)
        # per-fold weight: the stage-1 work of the fold plus the (fold-independent) stage-2 work, both in multiply-adds
        w = (w1 + float(nnz_y))[:total]
        out = torch.empty((args.folds, n), dtype=torch.float32, device="cuda")
        g.predict_loo(0, min(args.folds, total), clean=True, out=out[:min(args.folds, total)])   # warm: operands are cut at first use
        torch.cuda.synchronize()
        ok = True
    except Exception as e:
        ok = False
        state.setdefault("errors", []).append("c5 set-up on rank %d: %r" % (rank, e))
    if not agree(ok):
        return {"error": "set-up failed on a rank", "errors": state.get("errors")}

    def timed(lo, hi):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(lo, hi, args.folds):
            e = min(hi, b + args.folds)
            g.predict_loo(b, e, clean=True, out=out[:e - b])
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        per_rank = [mine]
        if world > 1:
            dev = "cuda" if backend == "nccl" else "cpu"
            tt = torch.zeros(world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(tt, torch.tensor([mine], dtype=torch.float64, device=dev))
            per_rank = [float(x) for x in tt.tolist()]
            tw = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            wall = float(tw.item())
        return wall, per_rank

    res = {"workload": "BASELINE configs[4]: 1e5 sources + 1e5 targets, Zipf(1.2) source degrees and target popularity, "
                       "nnz(Y) = %d (hottest source in %d targets, hottest target in %d sources), X as configs[2]; "
                       "leave-one-out over the first %d folds, sharded over %d rank(s), %d folds per launch"
                       % (nnz_y, ymax_row, ymax_col, total, world, args.folds),
           "folds_total": total}
    for name, wts in (("nnz_balanced_shards", w), ("equal_count_shards", None)):
        if name == "equal_count_shards" and world == 1:
            break   # one rank: both partitions are the whole range
        state["leg"] = "c5_loo:" + name
        lo, hi = ss.shard_range(total, rank, world, weights=wts)
        try:
            wall, per_rank = timed(lo, hi)
            good = True
        except Exception as e:
            good = False
            state.setdefault("errors", []).append("%s on rank %d: %r" % (name, rank, e))
        if not agree(good):
            res[name] = {"error": "a rank failed", "errors": state.get("errors")}
            break
        blocks = [ss.shard_range(total, r, world, weights=wts) for r in range(world)]
        loads = [float(w[a:b].sum()) for a, b in blocks]
        res[name] = {"wall_s": wall, "folds_per_s": total / wall, "edges_per_s": total * n / wall,
                     "folds_per_rank": [b - a for a, b in blocks], "seconds_per_rank": per_rank,
                     "work_imbalance_max_over_mean": max(loads) / (sum(loads) / world)}
    res["note"] = ("per-fold work at this config is nearly uniform (X is uniform; the power law sits in Y = the stage-2 operand, "
                   "which every fold multiplies in full and whose hot rows are split inside the kernel), so the two partitions "
                   "differ by a few folds; the weighted partition is what a skewed X needs")
    g.close()
    return res


def bench_c3loo(args, ss, torch, dist, world, rank, backend):
    """--workload c3loo: the configs[2] curve as the headline line (the default line carries it as `c3_loo`)."""
    res = c3_loo_curve(args, ss, torch, dist, world, rank, backend)
    if rank == 0:
        n = 100_000
        flops = 2.0 * res["nnz_Y"] * res["folds_per_rank_step"]
        by = csr_bytes(res["nnz_Y"], n) + n * res["folds_per_rank_step"] * 4 * 2
        print(json.dumps({
            "metric": "predicted edges/sec + achieved HBM GB/s, W*R SpMM", "value": res["edges_per_s"],
            "unit": "edges/s", "n_gpus": world, "steps": args.c3_steps, "warmup": 1,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": {"workload": res["workload"]},
            "roofline": {"kernel": "spmm_sell_kernel<float,4,true> (stage 2, B = %d folds)" % res["folds_per_rank_step"],
                         "bound": "hbm", "achieved": round(by / (res["stage2_ms"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(by / (res["stage2_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "avg_launch_ms": round(res["stage2_ms"], 4), "algorithmic_bytes": by,
                         "frac_fp32_fma": round(flops / (res["stage2_ms"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                         "note": "at B = thousands of columns the kernel is bound by LDS gathers / fp32 FMA issue, not by HBM"},
            "c3_loo": res}))
    if world > 1:
        dist.destroy_process_group()


def literal_dense_baseline(ss, n_small, dx, dy):
    """The reference path itself, literally (src/core.jl:365-371,402-423): dense N x N block adjacency A, B = A with the
    query rows/columns zeroed, W = spread(B) (row counts, Inf/NaN -> 0), F = A * (W * W), the queries x targets corner
    -- fp64 through numpy / OpenBLAS dgemm, the BLAS SimSpread.jl's `*` dispatches to.  Down-scaled C2 (N = 4 n_small
    nodes: the full C2 would need 12.8 GB per N x N array and 2.6e14 flop), same densities, same generator; the GPU
    scores the same down-scaled input for the error figure."""
    import scipy.sparse as sp
    Xq, Xs, Ys = synth_c2(n_small, n_small, n_small, n_small, dx, dy, seed=20250222 + 2, rank=0, weighted=True)
    q, s, f, t = n_small, n_small, n_small, n_small
    A = sp.bmat([[None, None, Xq, None], [None, None, Xs, Ys], [Xq.T, Xs.T, None, None], [None, Ys.T, None, None]],
                format="csr", dtype=np.float64)
    A = np.asarray(A.toarray(), dtype=np.float64)       # node order [queries; sources; features; targets] (src/core.jl:182-195)
    N = A.shape[0]
    assert N == q + s + f + t
    t0 = time.perf_counter()
    B = A.copy()                                        # deepcopy + zero the query rows/columns (src/core.jl:196-198)
    B[:q, :] = 0.0
    B[:, :q] = 0.0
    k = np.count_nonzero(B, axis=1).astype(np.float64)[:, None]   # k(G): non-zero count per row (src/graphs.jl:9-11)
    with np.errstate(divide="ignore", invalid="ignore"):
        W = B / k
    W[~np.isfinite(W)] = 0.0                            # Inf, NaN -> 0 (src/core.jl:367-368)
    F = A @ (W @ W)                                     # src/core.jl:413: A * W^2, power_by_squaring == W*W
    yhat = F[:q, q + s + f:]                            # F[names(ytest,1), names(ytest,2)] (src/core.jl:421)
    dt = time.perf_counter() - t0
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    got = g.predict("query")
    g.close()
    err = float(np.abs(got - yhat).max() / np.abs(yhat).max())
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [os.cpu_count()])
    except Exception:
        thr = os.cpu_count()
    return {"value": q * t / dt, "unit": "edges/s", "cores": int(thr), "kind": "port",
            "what": "literal dense restatement of src/core.jl:365-371,402-423 (A, B, spread, A*(W*W)) in fp64 on numpy/OpenBLAS dgemm "
                    "-- the BLAS the reference's `*` calls; Julia is not installed, so SimSpread.jl itself cannot be timed",
            "sample": "down-scaled C2: %d queries x %d targets, N = %d nodes (2 x N^3 x 2 = %.2e flop), one pass, %.1f s"
                      % (q, t, N, 4.0 * N ** 3, dt),
            "wall_s": dt, "gflops": 4.0 * N ** 3 / dt / 1e9, "max_rel_err_gpu_vs_cpu": err}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child and hand back its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed predict() passes (300 x ~2 ms: a >= 0.5 s timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--n", type=int, default=10_000, help="sources = features = targets")
    ap.add_argument("--dx", type=float, default=0.05)
    ap.add_argument("--dy", type=float, default=0.01)
    ap.add_argument("--unweighted", action="store_true", help="binary similarity features (featurize(..., weighted=false))")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3loo"],
                    help="c2: BASELINE configs[1] (default, the metric's config); c3loo: configs[2] as the headline line")
    ap.add_argument("--folds", type=int, default=2048, help="configs[2]: folds per rank and step")
    ap.add_argument("--c3-steps", type=int, default=5, help="configs[2]: timed steps")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="compute type (the metric's config is f32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the configs[2] leave-one-out curve")
    ap.add_argument("--no-c5", action="store_true", help="skip the configs[4] (power-law) nnz-balanced leave-one-out leg")
    ap.add_argument("--c5-folds", type=int, default=100_000, help="configs[4]: folds of the job that is sharded over the ranks")
    ap.add_argument("--dense-n", type=int, default=3000, help="literal dense CPU baseline: nodes per layer (N = 4x this)")
    ap.add_argument("--aux-timeout", type=int, default=300,
                    help="N > 1: seconds the auxiliary legs (configs[2] curve) may take before the headline line is printed without them")
    ap.add_argument("--gather", action="store_true", help="also time the final gather of the C2 score blocks (outside `value`)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, all-reduce one number over gloo on the CPU, print it and exit (tests the launch path without a GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process has touched the GPU
        # yet (no torch import, no HIP call), and the ranks are fresh child processes of torch.distributed.run -- never
        # an exec of a process that initialised the device.  Rank 0's JSON line goes to our stdout unchanged.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rendezvous_only:
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"rendezvous": {"world": world, "sum_of_rank_plus_1": int(t.item()), "gpus_arg": args.gpus}}), flush=True)
        return
    backend = os.environ.get("BENCH_BACKEND", "nccl")  # "gloo" + BENCH_SINGLE_DEVICE=1: rehearsal on a 1-GPU box
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)

    import simspread_jl_amd as ss
    ss.init(local_rank)
    ss.use_torch_stream()  # device buffers come from torch: share its stream

    if args.workload == "c3loo":
        return bench_c3loo(args, ss, torch, dist, world, rank, backend)

    nq, n = args.nq, args.n
    Xq, Xs, Ys = synth_c2(nq, n, n, n, args.dx, args.dy, seed=20250222 + 2, rank=rank, weighted=not args.unweighted)
    np_dt, th_dt, vb = (np.float32, torch.float32, 4) if args.dtype == "f32" else (np.float64, torch.float64, 8)
    g = ss.DeviceGraph.from_sparse(Xq.astype(np_dt), Xs.astype(np_dt), Ys.astype(np_dt), dtype=np_dt)
    scores = torch.empty((nq, n), dtype=th_dt, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        g.predict("query", out=scores)
    barrier()
    ss.timing_hold(True)   # HIP-event stage timings (on the stream the kernels run on) add up over the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.predict("query", out=scores)   # enqueued in stream order; the barrier below waits for all K steps
    barrier()
    elapsed = time.perf_counter() - t0
    t = ss.timing_last()
    ss.timing_hold(False)
    path = ss.path_last()
    # average duration of one launch of each kernel over the timed region (one launch per stage and step here)
    transfer_ms = t["transfer_ms"] / max(1, t["transfer_launches"])
    spmm_ms = t["spmm_ms"] / max(1, t["spmm_launches"])
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    ms_per_step = elapsed / args.steps * 1e3
    edges_per_step = nq * n * world
    value = edges_per_step / (elapsed / args.steps)

    gather_ms = None
    if args.gather and world > 1:
        src = scores if backend == "nccl" else scores.cpu()
        barrier()
        tg = time.perf_counter()
        ss.gather_scores(src, nq * world, counts=[nq] * world)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3

    import threading
    result = None
    state = {"printed": False, "leg": "headline"}
    emit_lock = threading.Lock()

    def emit(extra=None):
        # the ONE JSON line of the contract (rank 0), whatever happens to the auxiliary legs; the watchdog thread and the
        # main thread may both get here: first one prints, under the lock
        with emit_lock:
            if rank == 0 and result is not None and not state["printed"]:
                state["printed"] = True
                if extra:
                    result.update(extra)
                print(json.dumps(result), flush=True)

    def watchdog():
        # an auxiliary leg did not come back (a peer died inside an exchange, a hung collective): keep the headline, say
        # which leg hung, and make the hang visible in the exit code (never a silent rc 0)
        if rank != 0:
            time.sleep(5.0)   # let rank 0 print before the launcher tears the job down on the first non-zero exit
        emit({"aux_hang": {"leg": state["leg"], "timeout_s": args.aux_timeout, "errors": state.get("errors")}})
        sys.stderr.write("bench.py rank %d: auxiliary leg %r exceeded %d s; exiting 3\n" % (rank, state["leg"], args.aux_timeout))
        sys.stderr.flush()
        os._exit(3)

    if rank == 0:
        nnz_w = g.nnz_ys
        tname = "float" if args.dtype == "f32" else "double"
        # ---- dominant kernel of the step: stage 1 (sparse x sparse -> dense transfer block).  Algorithmic bytes per
        # launch: its inputs once (CSR(Xq) + CSR(Xs)) and its output once (T, nq x ns values); flops
        # 2 * nnz(Xq) * mean row length of Xs'.
        s1_bytes = csr_bytes(g.nnz_xq, nq, vb) + csr_bytes(g.nnz_xs, n, vb) + nq * n * vb
        s1_flops = 2.0 * g.nnz_xq * (g.nnz_xs / n)
        s1_pmc, s1_src = pmc_entry("transfer_kernel")
        s1_gbps = s1_bytes / (transfer_ms * 1e-3) / 1e9
        roofline = {
            "kernel": "transfer_kernel<%s> (stage 1, T = (Xq Df^-1) Xs' Ds^-1; %d%% of the step)"
                      % (tname, round(100 * transfer_ms / max(transfer_ms + spmm_ms, 1e-9))),
            "bound": "hbm", "achieved": round(s1_gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(s1_gbps / HBM_PEAK_GBS, 4), "frac_of_achievable_6p3TBps": round(s1_gbps / HBM_ACHIEVABLE_GBS, 4),
            "traffic": s1_pmc["hbm_bytes_per_launch"] if s1_pmc else None, "traffic_source": s1_src,
            "avg_launch_ms": round(transfer_ms, 4), "algorithmic_bytes": s1_bytes,
            "flops": s1_flops, "frac_fp32_fma": round(s1_flops / (transfer_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
            "limiter": "not HBM: two on-chip units, both measured (profiles/r03_stage1_mem_pmc.txt, DESIGN.md 4.1): (1) the vector memory "
                       "path of a CU moves ~28-32 B/clk whatever the load width (TA busy 92 %, TCP_TOTAL_CACHE_ACCESSES 0.78 per clk): every "
                       "query re-reads 5 % of X' as ~62-entry sub-rows from its XCD's L2 (95 % hits), ~18.5 GB per launch against "
                       "~17 TB/s for the chip; (2) the LDS scatter: one read-add-write pair per 64 entries at ~3 LDS cycles per 32-lane "
                       "group (the fullest of 32 banks holds ~6 of a sub-row's 62 columns), LDS 64 % busy.  Seven rebuilt variants of "
                       "round 3 (LDS-resident offsets, 16-byte loads, fixed-point ds_add_u32 sums, flat streams) all land at 1.8-3 ms",
            "l2_request_bytes": s1_pmc["l2_request_bytes_per_launch"] if s1_pmc else None,
            "frac_l2_sector_ceiling": (round(s1_pmc["l2_request_bytes_per_launch"] / (transfer_ms * 1e-3) / 1e9 / 18700.0, 4)
                                       if s1_pmc else None),
        }
        # ---- the W*R SpMM of the step (stage 2).  Algorithmic work per launch:
        #   bytes = CSR(W) + K*B*vb + M*B*vb,  flops = 2*nnz(W)*B,  B = nq columns of R per launch
        spmm_bytes = csr_bytes(nnz_w, n, vb) + n * nq * vb + n * nq * vb
        spmm_flops = 2.0 * nnz_w * nq
        s2_pmc, s2_src = pmc_entry("spmm_sell_kernel")
        achieved_tf = spmm_flops / (spmm_ms * 1e-3) / 1e12
        roofline_spmm = {
            "kernel": "spmm_sell_kernel<%s> (stage 2, F = W*R, B = %d)" % ("float,4" if args.dtype == "f32" else "double,2", nq),
            "bound": "fp32 vector FMA rate (no MFMA is issued; peak 157.3 TF = 256 CUs x 128 FMA/clk x 2.4 GHz); in practice the "
                     "LDS gather: one ds_read_b128 per four FMAs caps this formulation at 50 % of that peak",
            "achieved": round(achieved_tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved_tf / FP32_PEAK_TFLOPS, 4),
            "traffic": s2_pmc["hbm_bytes_per_launch"] if s2_pmc else None, "traffic_source": s2_src,
            "avg_launch_ms": round(spmm_ms, 4),
            "algorithmic_bytes": spmm_bytes,
            "algorithmic_GBps": round(spmm_bytes / (spmm_ms * 1e-3) / 1e9, 1),
            "frac_hbm": round(spmm_bytes / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        }
        result = {
            "metric": "predicted edges/sec + achieved HBM GB/s, W*R SpMM",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d queries x %d targets per GPU, %d sources/features, "
                                   "%.0f%% similarity (%s), %.0f%% bipartite density, fp32, full predict()"
                                   % (nq, n, n, args.dx * 100, "unweighted" if args.unweighted else "weighted U(0.5,1]",
                                      args.dy * 100),
                       "queries_per_gpu": nq, "sources": n, "features": n, "targets": n,
                       "nnz_Xq": g.nnz_xq, "nnz_Xs": g.nnz_xs, "nnz_Ys": g.nnz_ys, "sharding": "query rows, no collective",
                       "kernels": path, "timed_region_s": round(elapsed, 3),
                       "predict_algorithmic_bytes": csr_bytes(g.nnz_xq, nq) + csr_bytes(g.nnz_xs, n) + csr_bytes(nnz_w, n) + nq * n * 4},
            "roofline": roofline,
            "roofline_spmm": roofline_spmm,
        }
        if gather_ms is not None:
            result["score_gather_ms"] = gather_ms

    # ---- auxiliary legs, after the headline numbers are final and under a watchdog (never lose the headline line)
    timer = None
    if world > 1:
        timer = threading.Timer(args.aux_timeout, watchdog)
        timer.daemon = True
        timer.start()
    agree = make_agree(torch, dist, world, backend)
    if not args.no_c3:
        c3 = c3_loo_curve(args, ss, torch, dist, world, rank, backend, agree, state)
        if rank == 0:
            with emit_lock:
                result["c3_loo"] = c3
    if not args.no_c5:
        c5 = c5_loo_leg(args, ss, torch, dist, world, rank, backend, agree, state)
        if rank == 0:
            with emit_lock:
                result["c5_loo"] = c5
    state["leg"] = "done"
    if timer is not None:
        timer.cancel()   # the exchanges are over: what follows is rank-local (sweep, CPU baselines) and not under the timeout

    if rank == 0:
        if not args.no_sweep and world == 1:
            try:
                result["spmm_narrow_sweep"] = spmm_sweep(ss, torch)
            except Exception as e:  # the sweep is auxiliary; never lose the headline line
                result["spmm_narrow_sweep"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            from oracle import c_oracle
            f64 = [m.astype(np.float64) for m in (Xq, Xs, Ys)]
            # the graph-dependent part (transposes, reciprocal degrees) is prepared once, outside the timed body -- like
            # the GPU path, whose operands are resident when its timed region starts
            prep = c_oracle.Prepared(*f64)
            prep.predict(0, 8)  # warm
            nsample = min(nq, 512)
            tc = time.perf_counter()
            ref = prep.predict(0, nsample)
            first = time.perf_counter() - tc
            rows_per_pass = nq if first * nq / nsample < 20.0 else nsample
            buf = np.empty((rows_per_pass, n))
            rows_done, reps, dt = 0, 0, 0.0
            tc = time.perf_counter()
            while dt < 10.0 and reps < 50:   # bounded sample: whole passes over the same query block, ~10 s of CPU work
                ref = prep.predict(0, rows_per_pass, out=buf)
                rows_done += rows_per_pass
                reps += 1
                dt = time.perf_counter() - tc
            got = scores[:rows_per_pass].cpu().numpy()
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            prep.close()
            result["cpu_baseline"] = {
                "value": rows_done * n / dt, "unit": "edges/s", "cores": c_oracle.max_threads(), "kind": "port",
                "sample": "%d passes over the first %d of %d query rows of the same workload, fp64 CSR C/OpenMP "
                          "restatement of the reference algorithm in its factored form (oracle/factored.c, operands prepared "
                          "outside the timed body; Julia is not installed, so SimSpread.jl itself cannot be timed), %.1f s"
                          % (reps, rows_per_pass, nq, dt),
                "max_rel_err_gpu_vs_cpu": err,
            }
            try:
                result["cpu_baseline"]["literal_dense"] = literal_dense_baseline(ss, args.dense_n, args.dx, args.dy)
            except Exception as e:
                result["cpu_baseline"]["literal_dense"] = {"error": repr(e)}
    emit()
    if world > 1:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
