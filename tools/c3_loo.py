"""C3-shaped leave-one-out on one GPU: X symmetric-pattern n x n at `dens` + unit diagonal, Y n x n at `dens`,
a block of folds; prints stage timings and checks a few folds against the CPU oracle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss

def rand_sym_csr(n, dens, gen, diag=True):
    m = int(n * n * dens / 2)
    r = torch.randint(0, n, (m,), device="cuda", generator=gen, dtype=torch.int64)
    c = torch.randint(0, n, (m,), device="cuda", generator=gen, dtype=torch.int64)
    keys = torch.cat([r * n + c, c * n + r, torch.arange(n, device="cuda") * (n + 1)] if diag else [r * n + c, c * n + r])
    keys = torch.unique(keys)
    rr = torch.div(keys, n, rounding_mode="floor")
    idx = (keys - rr * n).to(torch.int32)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    ptr[1:] = torch.cumsum(torch.bincount(rr, minlength=n), 0)
    return ptr, idx

def rand_csr(n, m_cols, dens, gen):
    m = int(n * m_cols * dens)
    r = torch.randint(0, n, (m,), device="cuda", generator=gen, dtype=torch.int64)
    c = torch.randint(0, m_cols, (m,), device="cuda", generator=gen, dtype=torch.int64)
    keys = torch.unique(r * m_cols + c)
    rr = torch.div(keys, m_cols, rounding_mode="floor")
    idx = (keys - rr * m_cols).to(torch.int32)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    ptr[1:] = torch.cumsum(torch.bincount(rr, minlength=n), 0)
    return ptr, idx

def main():
    n = int(os.environ.get("N", 100_000)); dens = float(os.environ.get("DENS", 0.01)); folds = int(os.environ.get("FOLDS", 2048))
    ss.init(0); ss.use_torch_stream()
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 3)
    xp, xi = rand_sym_csr(n, dens, gen)
    yp, yi = rand_csr(n, n, dens, gen)
    xv = (0.5 + 0.5 * torch.rand(xi.numel(), device="cuda", generator=gen)).float()
    t0 = time.perf_counter()
    g = ss.DeviceGraph.from_device_csr(0, n, n, n, None, (xp, xi, xv), (yp, yi, None), dtype=np.float32)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    out = torch.empty((folds, n), dtype=torch.float32, device="cuda")
    res = {}
    for it in range(3):
        t0 = time.perf_counter()
        g.predict_loo(0, folds, clean=True, out=out)
        torch.cuda.synchronize()
        res = dict(ss.timing_last(), wall_ms=(time.perf_counter() - t0) * 1e3)
    print(json.dumps(dict(n=n, nnz_x=int(xi.numel()), nnz_y=int(yi.numel()), folds=folds, build_s=t_build, **res,
                          edges_per_s=folds * n / (res["wall_ms"] * 1e-3))))
    if os.environ.get("CHECK", "1") == "1":
        import scipy.sparse as sp
        from oracle import simspread_oracle as O
        X = sp.csr_matrix((xv.cpu().numpy().astype(np.float64), xi.cpu().numpy(), xp.cpu().numpy()), shape=(n, n))
        Y = sp.csr_matrix((np.ones(yi.numel()), yi.cpu().numpy(), yp.cpu().numpy()), shape=(n, n))
        qs = [0, folds // 2, folds - 1]
        want = O.predict_loo_factored(X, Y, clean_flag=True, queries=qs)
        got = out[qs].cpu().numpy().astype(np.float64)
        err = np.abs(got - want).max() / np.abs(want).max()
        print("max rel err vs oracle on folds", qs, ":", err)
        assert err < 1e-5
if __name__ == "__main__":
    main()
