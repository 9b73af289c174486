"""Randomised cross-check of ss_spmm_* (every width route, both precisions, pattern-only and weighted, odd shapes, forced
small chunks) against scipy, and of ss_topl_f32 against a stable descending sort.  python tools/fuzz_spmm_topl.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import simspread_jl_amd as ss

def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    ss.init(0)
    rng = np.random.default_rng(int(os.environ.get("SEED", 12345)))
    worst = 0.0
    for case in range(ncases):
        M = int(rng.integers(1, 4000)); K = int(rng.integers(1, 9000))
        dens = float(rng.choice([0.0005, 0.003, 0.02, 0.08, 0.3]))
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        W = sp.random(M, K, density=dens, format="csr", random_state=rng, dtype=np.float64)
        if rng.random() < 0.3:
            W.data[:] = 1.0
        else:
            W.data = rng.random(W.nnz) + 0.5
        if rng.random() < 0.3 and M > 3:       # a few heavy rows and empty rows
            W = W.tolil(); W[int(rng.integers(0, M)), :] = 1.0; W[int(rng.integers(0, M)), :] = 0.0; W = W.tocsr()
        if rng.random() < 0.5:
            os.environ["SS_NARROW_CHUNK"] = str(int(rng.choice([64, 100, 333, 1000])))
            os.environ["SS_SELL_CHUNK"] = str(int(rng.choice([64, 500, 2000])))
        else:
            os.environ.pop("SS_NARROW_CHUNK", None); os.environ.pop("SS_SELL_CHUNK", None)
        w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
        for B in rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 130], size=4, replace=False):
            B = int(B)
            R = rng.standard_normal((K, B))
            got = np.asarray(w.spmm(R.astype(dtype)), np.float64)
            want = W @ R
            scale = max(np.abs(W).dot(np.abs(R)).max(), 1e-300)
            err = np.abs(got - want).max() / scale
            tol = 2e-6 if dtype == np.float32 else 1e-13
            worst = max(worst, err / tol)
            assert err <= tol, (case, M, K, dens, dtype, B, err, ss.path_last())
        w.close() if hasattr(w, "close") else None
    print("spmm: %d cases ok, worst error / tolerance %.3f" % (ncases, worst))
    for case in range(ncases // 3):
        rows = int(rng.integers(1, 40)); n = int(rng.integers(1, 120000))
        x = rng.standard_normal((rows, n)).astype(np.float32)
        mode = rng.integers(0, 4)
        if mode == 1: x = np.round(x * 3) / 3
        if mode == 2: x[:, rng.random(n) < 0.9] = 0.0
        if mode == 3: x[rng.integers(0, rows)] = -99.0
        L = int(min(n, rng.choice([1, 5, 100, 1000, 1024])))
        idx, val = ss.topl(x, L)
        # sortperm(x, rev=true) as Julia orders floats (isless): score descending, +0.0 before -0.0, ties by ascending column
        # (numpy's argsort(-x) calls +0.0 and -0.0 equal: seed 7 found that difference, it is the test's, not the kernel's)
        u = x.view(np.uint32)
        key = np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint64)
        comp = (key << np.uint64(32)) | (~np.arange(n, dtype=np.uint32)).astype(np.uint64)
        want = np.argsort(comp, axis=1)[:, ::-1][:, :L]
        assert np.array_equal(idx, want), (case, rows, n, L, mode)
        assert np.array_equal(val, np.take_along_axis(x, want, 1))
    print("topl: %d cases ok" % (ncases // 3))

if __name__ == "__main__":
    main()
