"""Time of a 10-fold cross-validation (ss_predict_kfold_f32: per-fold degree recount + predict of the fold's members) on a
C2-shaped LOO graph: 10k sources, symmetric 5 % similarity, 1 % labels; host output (10k x 10k scores)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import simspread_jl_amd as ss

ss.init(0)
n = int(os.environ.get("N", 10000)); k = int(os.environ.get("K", 10))
rng = np.random.default_rng(7)
X = sp.random(n, n, density=0.025, format="csr", random_state=rng, dtype=np.float32)
X = X + X.T; X.setdiag(1.0); X = sp.csr_matrix(X); X.data = (0.5 + 0.5 * rng.random(X.nnz)).astype(np.float32)
Y = sp.random(n, n, density=0.01, format="csr", random_state=rng, dtype=np.float32); Y.data[:] = 1.0
g = ss.DeviceGraph.from_sparse(None, X, Y, dtype=np.float32)
fold = rng.integers(0, k, n).astype(np.int32)
for rep in range(3):
    t0 = time.perf_counter()
    out = g.predict_kfold(fold, k, clean=True)
    t1 = time.perf_counter()
    t = ss.timing_last()
    print(json.dumps({"n": n, "folds": k, "wall_ms": round((t1 - t0) * 1e3, 2), "transfer_ms": round(t["transfer_ms"], 3),
                      "spmm_ms": round(t["spmm_ms"], 3), "epilogue_ms": round(t["epilogue_ms"], 3), "d2h_ms": round(t["d2h_ms"], 3),
                      "total_device_ms": round(t["total_ms"], 3)}))
