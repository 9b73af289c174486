"""Randomised cross-check of the whole predict path against the CPU oracle: random tri-partite graphs (sizes, densities,
weighted / pattern-only, empty rows, zero-degree features and targets, a hot target), both precisions, every row kind
(query rows, source rows, leave-one-out blocks, k-fold), clean! on and off, and a random draw of the library's tuning
switches (stage-1 kernels of round 3, chunk sizes, sorted SELL operand).  python tools/fuzz_predict.py [seconds] [seed]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import simspread_jl_amd as ss
from oracle import simspread_oracle as O

SWITCHES = {
    "SS_TRANSFER_V": [None, None, "1", "2"],
    "SS_TRANSFER_FIX": [None, "0"],
    "SS_TRANSFER_QFLAT": [None, None, "4"],
    "SS_TRANSFER_WIDE": [None, None, "4", "2"],
    "SS_TRANSFER_WIDE2": [None, None, "1"],
    "SS_TRANSFER_FIX1": [None, None, "1"],
    "SS_TRANSFER_LD": [None, None, "1"],
    "SS_TRANSFER_U": [None, "4", "8"],
    "SS_TRANSFER_CHUNK": [None, None, "64", "500", "3000"],
    "SS_SELL_CHUNK": [None, None, "64", "777"],
    "SS_SELL_SORT": [None, None, "0", "1"],
    "SS_CHUNK_SCHED": [None, None, "0"],
    "SS_TRANSFER_BYTES": [None, None, str(1 << 20)],
}


def rand_graph(rng, dtype):
    nq = int(rng.integers(1, 700))
    ns = int(rng.choice([3, 17, 64, 200, 1000, 3000, 6000]))
    nf = ns if rng.random() < 0.7 else int(rng.integers(1, 2 * ns + 2))
    nt = int(rng.choice([1, 2, 31, 64, 257, 1000]))
    dx = float(rng.choice([0.002, 0.02, 0.05, 0.2, 0.6]))
    dy = float(rng.choice([0.005, 0.03, 0.2]))
    weighted = rng.random() < 0.6

    def mat(r, c, d, w):
        m = sp.random(r, c, density=d, format="csr", random_state=rng, dtype=np.float64)
        m.data = (0.05 + rng.random(m.nnz)) if w else np.ones(m.nnz)
        return m

    Xq, Xs, Ys = mat(nq, nf, dx, weighted), mat(ns, nf, dx, weighted), mat(ns, nt, dy, False)
    Xq, Xs, Ys = Xq.tolil(), Xs.tolil(), Ys.tolil()
    if nq > 2:
        Xq[int(rng.integers(0, nq)), :] = 0                          # a query without features
    if nf > 3:
        Xs[:, int(rng.integers(0, nf))] = 0                          # a feature nobody has
    if nt > 2:
        Ys[:, int(rng.integers(0, nt))] = 0                          # a target without edges (clean!)
        Ys[:, int(rng.integers(0, nt))] = 1                          # a target every source has
    if ns > 4:
        Ys[int(rng.integers(0, ns)), :] = 0                          # an isolated source
    conv = lambda m: sp.csr_matrix(m).astype(dtype)
    return conv(Xq), conv(Xs), conv(Ys), weighted


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20250222
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None   # replay one case of a seed (the draws before it are repeated, not scored)
    rng = np.random.default_rng(seed)
    ss.init(0)
    t0 = last = time.time()
    cases, worst32, worst64, kinds, paths = 0, 0.0, 0.0, {}, {}
    while time.time() - t0 < budget and (only is None or cases <= only):
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        env = {}
        for k, choices in SWITCHES.items():
            v = choices[int(rng.integers(0, len(choices)))]
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
                env[k] = v
        Xq, Xs, Ys, weighted = rand_graph(rng, dtype)
        f64 = [m.astype(np.float64) for m in (Xq, Xs, Ys)]
        if only is not None and cases != only:
            rng.random()
            if Xs.shape[0] == Xs.shape[1] and rng.random() < 0.7:
                lo = int(rng.integers(0, Xs.shape[0])); int(rng.integers(lo, Xs.shape[0]))
                if Xs.shape[0] <= 3000:
                    rng.integers(0, Xs.shape[0], size=6)
            cases += 1
            continue
        if only is not None:
            print("replaying", dict(env=env, shapes=(Xq.shape, Xs.shape, Ys.shape), nnz=(Xq.nnz, Xs.nnz, Ys.nnz), weighted=weighted, dtype=str(dtype)), flush=True)
            if os.environ.get("FUZZ_SAVE"):
                sp.save_npz(os.environ["FUZZ_SAVE"] + "_xq.npz", Xq); sp.save_npz(os.environ["FUZZ_SAVE"] + "_xs.npz", Xs); sp.save_npz(os.environ["FUZZ_SAVE"] + "_ys.npz", Ys)
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=dtype)
        tol = 1e-5 if dtype == np.float32 else 1e-12
        kt = np.asarray((f64[2] != 0).sum(0)).ravel()
        checks = []
        clean = bool(rng.random() < 0.5)
        want = O.predict_factored(*f64, rows="query")
        if clean:
            want = want.copy(); want[:, kt == 0] = -99.0
        checks.append(("query", g.predict("query", clean=clean), want))
        for p in ss.path_last():
            paths[p] = paths.get(p, 0) + 1
        if Xs.shape[0] == Xs.shape[1] and rng.random() < 0.7:
            ws = O.predict_factored(None, f64[1], f64[2], rows="source")
            lo = int(rng.integers(0, Xs.shape[0])); hi = int(rng.integers(lo, Xs.shape[0])) + 1
            checks.append(("source", g.predict("source", lo, hi), ws[lo:hi]))
            if Xs.shape[0] <= 3000:
                # leave-one-out lives on the 3-layer graph (no query rows; feature j is named after source j)
                g3 = ss.DeviceGraph.from_sparse(None, Xs, Ys, dtype=dtype)
                qs = sorted(set(int(x) for x in rng.integers(0, Xs.shape[0], size=6)))
                wl = O.predict_loo_factored(f64[1], f64[2], clean_flag=clean, queries=qs)
                lo, hi = min(qs), max(qs) + 1
                gl = g3.predict_loo(lo, hi, clean=clean)
                for p in ss.path_last():
                    paths[p] = paths.get(p, 0) + 1
                checks.append(("loo", gl[[q - lo for q in qs]].copy(), wl))
                g3.close()
        for name, got, w in checks:
            scale = max(np.abs(w[w != -99.0]).max() if (w != -99.0).any() else 0.0, 1e-300)
            err = np.abs(np.asarray(got, np.float64) - w).max() / scale
            assert err <= tol, dict(kind=name, err=err, env=env, shape=(Xq.shape, Xs.shape, Ys.shape), weighted=weighted,
                                    dtype=str(dtype), path=ss.path_last(), seed=seed, case=cases)
            assert ((w == -99.0) == (np.asarray(got) == -99.0)).all()
            if dtype == np.float32:
                worst32 = max(worst32, err)
            else:
                worst64 = max(worst64, err)
            kinds[name] = kinds.get(name, 0) + 1
        g.close()
        cases += 1
        if time.time() - last > 45:
            last = time.time()
            print(f"[{last - t0:.0f} s] {cases} graphs ok", flush=True)
    for k in SWITCHES:
        os.environ.pop(k, None)
    print(json.dumps({"seconds": round(time.time() - t0, 1), "seed": seed, "graphs": cases, "checks": kinds, "kernels_taken": paths,
                      "worst_rel_err_fp32": worst32, "worst_rel_err_fp64": worst64,
                      "source_sha": ss._lib.source_hash()}))


if __name__ == "__main__":
    main()
