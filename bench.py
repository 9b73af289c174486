#!/usr/bin/env python3
"""Benchmark of the SimSpread hot path on MI355X.

One "step" = one full predict() pass (stage 1 transfer block + stage 2 W*R SpMM) over one batch of
synthetic input that is already resident in HBM: BASELINE.json configs[1]
    10k queries x 10k targets, 10k sources/features, 5 % similarity, 1 % bipartite density, fp32.
Multi-GPU (torch.distributed, one rank per GPU): query rows are independent, so every rank scores its
own 10k-query shard against the replicated source/target graph (weak scaling, no data-path collective).

Prints ONE JSON line on rank 0 (driver contract) with `roofline` and `cpu_baseline` objects.
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


def pmc_field(kernel_substr, field):
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        for k, v in d["kernels"].items():
            if kernel_substr in k:
                return v.get(field)
    except Exception:
        pass
    return None


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/pmc_latest.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for wide coalesced reads on gfx950).  bench.py cannot run the profiler on itself, so this
    is the figure measured for the same command when the profile was taken; null if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        for k, v in d["kernels"].items():
            if kernel_substr in k:
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def csr_bytes(nnz, rows, vb=4):
    return nnz * (vb + 4) + (rows + 1) * 4


def spmm_sweep(ss, torch, steps=5):
    """Narrow-R regime of the W*R SpMM at the north-star size: W 100k x 100k, 1 % (nnz ~1e8), fp32,
    B in {1,4,8,16,32,64}; HBM roofline fraction on algorithmic bytes."""
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
    widths = tuple(int(x) for x in os.environ.get("SWEEP_B", "1,4,8,16,32,64").split(","))
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
        out.append({"B": B, "ms": round(t * 1e3, 4), "GBps": round(by / t / 1e9, 1),
                    "frac_hbm": round(by / t / 1e9 / HBM_PEAK_GBS, 4), "bytes": by})
    lib.ss_spmat_destroy(h)
    return {"workload": f"W 100k x 100k, 1 percent dense (nnz {nnz}), fp32, CSR streamed once from HBM", "results": out}


def bench_c3loo(args, ss, torch, dist, world, rank):
    """BASELINE configs[2]: 100k x 100k, 1 %, leave-one-out; folds are block-sharded over the ranks
    (ss.shard_range) and each step scores `--folds` consecutive folds of the rank's shard."""
    from tools.c3_loo import rand_csr, rand_sym_csr
    n = 100_000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250222 + 3)   # every rank builds the same (replicated) graph
    xp, xi = rand_sym_csr(n, 0.01, gen)
    yp, yi = rand_csr(n, n, 0.01, gen)
    xv = (0.5 + 0.5 * torch.rand(xi.numel(), device="cuda", generator=gen)).float()
    g = ss.DeviceGraph.from_device_csr(0, n, n, n, None, (xp, xi, xv), (yp, yi, None), dtype=np.float32)
    lo, hi = ss.shard_range(n, rank, world)
    folds = min(args.folds, hi - lo)
    out = torch.empty((folds, n), dtype=torch.float32, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pos = lo
    for _ in range(args.warmup):
        g.predict_loo(pos, pos + folds, clean=True, out=out)
    barrier()
    ss.timing_hold(True)   # HIP-event stage timings add up over the timed region, read once after it
    t0 = time.perf_counter()
    for i in range(args.steps):
        b = lo + (i * folds) % max(1, (hi - lo) - folds + 1)
        g.predict_loo(b, b + folds, clean=True, out=out)
    barrier()
    elapsed = time.perf_counter() - t0
    t = ss.timing_last()
    ss.timing_hold(False)
    st = {"transfer_ms": [t["transfer_ms"] / max(1, t["transfer_launches"])],
          "spmm_ms": [t["spmm_ms"] / max(1, t["spmm_launches"])]}
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if os.environ.get("BENCH_BACKEND", "nccl") == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    if rank == 0:
        spmm_ms = float(np.mean(st["spmm_ms"]))
        flops = 2.0 * g.nnz_ys * folds
        by = csr_bytes(g.nnz_ys, n) + n * folds * 4 * 2
        print(json.dumps({
            "metric": "predicted edges/sec + achieved HBM GB/s, W*R SpMM", "value": folds * n * world / (elapsed / args.steps),
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 100k x 100k, 1%% density, leave-one-out folds block-sharded over "
                                   "ranks, %d folds per rank and step" % folds, "nnz_X": g.nnz_xs, "nnz_Y": g.nnz_ys,
                       "folds_per_s_per_gpu": folds / (elapsed / args.steps)},
            "roofline": {"kernel": "spmm_sell_kernel<float,4> (stage 2, B = %d folds, 10 LDS chunks of W)" % folds,
                         "bound": "mfma", "achieved": round(flops / (spmm_ms * 1e-3) / 1e12, 3), "peak": FP32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(flops / (spmm_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                         "traffic": None, "avg_launch_ms": round(spmm_ms, 4), "algorithmic_bytes": by,
                         "frac_hbm": round(by / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "stage1_transfer_ms": round(float(np.mean(st["transfer_ms"])), 4)}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--n", type=int, default=10_000, help="sources = features = targets")
    ap.add_argument("--dx", type=float, default=0.05)
    ap.add_argument("--dy", type=float, default=0.01)
    ap.add_argument("--unweighted", action="store_true", help="binary similarity features (featurize(..., weighted=false))")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3loo"],
                    help="c2: BASELINE configs[1] (default, the metric's config); c3loo: configs[2], 100k x 100k 1%% "
                         "leave-one-out, each rank scores --folds consecutive folds of its shard per step")
    ap.add_argument("--folds", type=int, default=2048, help="c3loo: folds per rank and step")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="compute type (the metric's config is f32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time an RCCL all_gather of the score blocks (outside `value`)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("BENCH_BACKEND", "nccl")  # "gloo" + BENCH_SINGLE_DEVICE=1: rehearsal on a 1-GPU box
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    elif args.gpus > 1:
        print("bench.py: --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)

    import simspread_jl_amd as ss
    ss.init(local_rank)
    ss.use_torch_stream()  # device buffers come from torch: share its stream

    if args.workload == "c3loo":
        return bench_c3loo(args, ss, torch, dist, world, rank)

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
    ss.timing_hold(True)   # HIP-event stage timings add up over the timed region, read once after it
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.predict("query", out=scores)   # enqueued in stream order; the barrier below waits for all K steps
    barrier()
    elapsed = time.perf_counter() - t0
    t = ss.timing_last()
    ss.timing_hold(False)
    # average duration of one launch of each kernel over the timed region (one launch per stage and step here)
    stage = {"transfer_ms": [t["transfer_ms"] / max(1, t["transfer_launches"])],
             "spmm_ms": [t["spmm_ms"] / max(1, t["spmm_launches"])], "total_ms": [t["total_ms"] / args.steps]}
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    ms_per_step = elapsed / args.steps * 1e3
    edges_per_step = nq * n * world
    value = edges_per_step / (elapsed / args.steps)

    gather_ms = None
    if args.gather and world > 1:
        bufs = [torch.empty_like(scores) for _ in range(world)]
        barrier()
        tg = time.perf_counter()
        dist.all_gather(bufs, scores)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3

    if rank == 0:
        nnz_w = g.nnz_ys
        spmm_ms = float(np.mean(stage["spmm_ms"]))
        transfer_ms = float(np.mean(stage["transfer_ms"]))
        # dominant kernel: the W*R SpMM (stage 2).  Algorithmic work per launch:
        #   bytes = CSR(W) + K*B*4 + M*B*4,  flops = 2*nnz(W)*B,  B = nq columns of R per launch
        spmm_bytes = csr_bytes(nnz_w, n, vb) + n * nq * vb + n * nq * vb
        spmm_flops = 2.0 * nnz_w * nq
        achieved_tf = spmm_flops / (spmm_ms * 1e-3) / 1e12
        roofline = {
            "kernel": "spmm_sell_kernel<%s> (stage 2, F = W*R, B = %d)" % ("float,4" if args.dtype == "f32" else "double,2", nq),
            "bound": "mfma",
            "bound_note": "wide-R SpMM is FMA/LDS-gather bound; peak = fp32 vector rate = fp32-input MFMA rate (157.3 TF)",
            "achieved": round(achieved_tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved_tf / FP32_PEAK_TFLOPS, 4),
            "traffic": pmc_traffic("spmm_sell_kernel"),
            "avg_launch_ms": round(spmm_ms, 4),
            "algorithmic_bytes": spmm_bytes,
            "algorithmic_GBps": round(spmm_bytes / (spmm_ms * 1e-3) / 1e9, 1),
            "frac_hbm": round(spmm_bytes / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "predict_algorithmic_bytes": csr_bytes(g.nnz_xq, nq) + csr_bytes(g.nnz_xs, n) + csr_bytes(nnz_w, n) + nq * n * 4,
        }
        # stage 1 (sparse x sparse -> dense transfer block) is the longer kernel at this shape; it is bound by
        # LDS scatter throughput (73 % LDS-busy in profiles/), not by HBM or FMA -- both fractions are reported
        s1_bytes = csr_bytes(g.nnz_xq, nq) + csr_bytes(g.nnz_xs, n) + nq * n * 4
        s1_flops = 2.0 * g.nnz_xq * (g.nnz_xs / n)
        stage1 = {
            "kernel": "transfer_kernel<float,false,8> (stage 1, T = (Xq Df^-1) Xs' Ds^-1)",
            "avg_launch_ms": round(transfer_ms, 4),
            "bound": "L2 bandwidth for short runs (DESIGN.md 4.1): every query re-reads 5 % of X' as ~62-entry sub-rows; "
                     "tools/subrow_fetch_bench.hip measures 18.7 TB/s as the chip's ceiling for that pattern",
            "l2_bytes": pmc_field("transfer_kernel", "l2_request_bytes_per_launch"),
            "l2_peak_GBps_measured": 18700.0,
            "frac_l2": (round(pmc_field("transfer_kernel", "l2_request_bytes_per_launch") / (transfer_ms * 1e-3) / 1e9 / 18700.0, 4)
                        if pmc_field("transfer_kernel", "l2_request_bytes_per_launch") else None),
            "algorithmic_bytes": s1_bytes, "frac_hbm": round(s1_bytes / (transfer_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "flops": s1_flops, "frac_fma": round(s1_flops / (transfer_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
            "traffic": pmc_traffic("transfer_kernel"),
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
                       "nnz_Xq": g.nnz_xq, "nnz_Xs": g.nnz_xs, "nnz_Ys": g.nnz_ys, "sharding": "query rows, no collective"},
            "roofline": roofline,
            "roofline_stage1": stage1,
        }
        if gather_ms is not None:
            result["score_gather_ms"] = gather_ms
        if not args.no_sweep and world == 1:
            try:
                result["spmm_narrow_sweep"] = spmm_sweep(ss, torch)
            except Exception as e:  # the sweep is auxiliary; never lose the headline line
                result["spmm_narrow_sweep"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            from oracle import c_oracle
            f64 = [m.astype(np.float64) for m in (Xq, Xs, Ys)]
            c_oracle.predict_query(*f64, r0=0, r1=8)  # warm
            # bounded sample: whole passes over the same query block until ~10 s of CPU work are done
            nsample, reps, dt = min(nq, 512), 0, 0.0
            tc = time.perf_counter()
            ref = c_oracle.predict_query(*f64, r0=0, r1=nsample)
            first = time.perf_counter() - tc
            rows_per_pass = nq if first * nq / nsample < 20.0 else nsample
            rows_done = 0
            tc = time.perf_counter()
            while dt < 10.0 and reps < 50:
                ref = c_oracle.predict_query(*f64, r0=0, r1=rows_per_pass)
                rows_done += rows_per_pass
                reps += 1
                dt = time.perf_counter() - tc
            got = scores[:rows_per_pass].cpu().numpy()
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            result["cpu_baseline"] = {
                "value": rows_done * n / dt, "unit": "edges/s", "cores": c_oracle.max_threads(), "kind": "port",
                "sample": "%d passes over the first %d of %d query rows of the same workload, fp64 CSR C/OpenMP "
                          "restatement of the reference algorithm (oracle/factored.c; Julia is not installed, so "
                          "SimSpread.jl itself cannot be timed), %.1f s" % (reps, rows_per_pass, nq, dt),
                "max_rel_err_gpu_vs_cpu": err,
            }
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
