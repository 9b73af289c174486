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

def zipf_degrees(n, nt, mean_deg, alpha):
    """Row degree of source rank r proportional to r^-alpha, capped at nt, scaled so that the CAPPED degrees have the
    requested mean ('normalised to mean 1 000 (nnz ~1e8, max row capped at Nt)')."""
    w = torch.arange(1, n + 1, device="cuda", dtype=torch.float64) ** (-alpha)
    scale = mean_deg * n / w.sum()
    for _ in range(60):  # fixed point: the capped rows lose mass, the others are scaled up to make up for it
        deg = torch.clamp(w * scale, max=float(nt))
        scale = scale * (mean_deg * n) / deg.sum()
    return torch.clamp(w * scale, 1.0, float(nt))


def zipf_bipartite_spec(n, nt, mean_deg, alpha, gen, block=1000):
    """C5 at its specified weight: nnz(Y) ~ n * mean_deg AFTER de-duplication.  Poisson (independent inclusion)
    sampling: edge (r, t) is drawn with probability min(1, deg_r * p_t * lam_r), p_t proportional to t^-alpha, lam_r
    chosen per row so that the expected row degree is deg_r (rows that saturate their hot targets spread the rest
    over the colder ones).  No duplicates by construction, so nothing is lost afterwards; hot sources and hot
    targets sit at random positions.  Returns CSR (ptr int64, idx int32) on the device, rows sorted."""
    deg = zipf_degrees(n, nt, mean_deg, alpha)
    deg = deg[torch.randperm(n, device="cuda", generator=gen)]
    pt = torch.arange(1, nt + 1, device="cuda", dtype=torch.float64) ** (-alpha)
    pt = (pt / pt.sum())[torch.randperm(nt, device="cuda", generator=gen)].float()
    counts = torch.zeros(n, dtype=torch.int64, device="cuda")
    cols = []
    for r0 in range(0, n, block):
        d = deg[r0:r0 + block].float()[:, None]
        lam = torch.ones_like(d)
        for _ in range(12):   # water-filling: sum_t min(1, d * p_t * lam) == d
            P = torch.clamp(d * pt[None, :] * lam, max=1.0)
            lam = lam * d / torch.clamp(P.sum(dim=1, keepdim=True), min=1e-30)
        P = torch.clamp(d * pt[None, :] * lam, max=1.0)
        hit = torch.rand(P.shape, device="cuda", generator=gen) < P
        rr, cc = torch.nonzero(hit, as_tuple=True)      # row-major order: rows ascending, columns ascending
        counts[r0:r0 + block] = torch.bincount(rr, minlength=P.shape[0])
        cols.append(cc.to(torch.int32))
        del P, hit, rr, cc
    ptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr, torch.cat(cols)


def main():
    n = int(os.environ.get("N", 100_000)); folds = int(os.environ.get("FOLDS", 2048)); mean_deg = float(os.environ.get("MEAN_DEG", 1000))
    dens = float(os.environ.get("DENS", 0.01))
    ss.init(0); ss.use_torch_stream()
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 5)
    xp, xi = rand_sym_csr(n, dens, gen)
    yp, yi = (zipf_bipartite_spec if os.environ.get("SPEC", "1") == "1" else zipf_bipartite)(n, n, mean_deg, 1.2, gen)
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
