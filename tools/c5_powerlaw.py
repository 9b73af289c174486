"""C5-shaped (power-law) leave-one-out block on one GPU: source degrees and target popularity Zipf(1.2),
hot rows/columns scattered at random positions; X symmetric-pattern uniform as in C3."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss
from tools.c3_loo import rand_sym_csr

def zipf_bipartite(n, nt, mean_deg, alpha, gen):
    ranks = torch.arange(1, n + 1, device="cuda", dtype=torch.float64)
    deg = ranks ** (-alpha)
    deg = torch.clamp((deg / deg.mean() * mean_deg).round().long(), 1, nt)
    deg = deg[torch.randperm(n, device="cuda", generator=gen)]          # hot sources anywhere
    pt = torch.arange(1, nt + 1, device="cuda", dtype=torch.float64) ** (-alpha)
    total = int(deg.sum().item())
    cols = torch.multinomial((pt / pt.sum()).float(), total, replacement=True, generator=gen)
    cols = torch.randperm(nt, device="cuda", generator=gen)[cols]       # hot targets anywhere
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    keys = torch.unique(rows * nt + cols)
    rr = torch.div(keys, nt, rounding_mode="floor")
    idx = (keys - rr * nt).to(torch.int32)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    ptr[1:] = torch.cumsum(torch.bincount(rr, minlength=n), 0)
    return ptr, idx

def main():
    n = int(os.environ.get("N", 100_000)); folds = int(os.environ.get("FOLDS", 2048)); mean_deg = float(os.environ.get("MEAN_DEG", 1000))
    dens = float(os.environ.get("DENS", 0.01))
    ss.init(0); ss.use_torch_stream()
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 5)
    xp, xi = rand_sym_csr(n, dens, gen)
    yp, yi = zipf_bipartite(n, n, mean_deg, 1.2, gen)
    xv = (0.5 + 0.5 * torch.rand(xi.numel(), device="cuda", generator=gen)).float()
    colcnt = torch.bincount(yi.long(), minlength=n)
    g = ss.DeviceGraph.from_device_csr(0, n, n, n, None, (xp, xi, xv), (yp, yi, None), dtype=np.float32)
    out = torch.empty((folds, n), dtype=torch.float32, device="cuda")
    res = {}
    for it in range(3):
        t0 = time.perf_counter()
        g.predict_loo(0, folds, clean=True, out=out)
        torch.cuda.synchronize()
        res = dict(ss.timing_last(), wall_ms=(time.perf_counter() - t0) * 1e3)
    print(json.dumps(dict(n=n, nnz_x=int(xi.numel()), nnz_y=int(yi.numel()), max_target_degree=int(colcnt.max().item()),
                          max_source_degree=int((yp[1:] - yp[:-1]).max().item()), folds=folds, **res,
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
        print("max rel err vs oracle on folds", qs, ":", err, " clean flags:", int((want == -99).sum()), int((got == -99).sum()))
        assert err < 1e-5 and ((want == -99) == (got == -99)).all()
if __name__ == "__main__":
    main()
