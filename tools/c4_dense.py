"""C4-shaped dense-similarity run on one GPU: S n x n fp32 symmetric U(0,1), unit diagonal; Y n x nt at 1 %;
cutoff sweep alpha in {0.1 .. 0.9} (fill 90 % .. 10 %); a block of leave-one-out folds per alpha.
Reports stage-1 (MFMA GEMM with fused cutoff) TFLOP/s = 2*folds*n*n / time."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss
from tools.c3_loo import rand_csr

def sym_uniform(n, gen):
    """Raw similarity of SURVEY.md 8d (C4): symmetric, iid U(0,1) off the diagonal, unit diagonal, so that
    fill(S >= alpha) = 1 - alpha.  (The mean of two uniforms, (R + R')/2, is triangular: 98 % fill at alpha = 0.1.)"""
    S = torch.rand((n, n), device="cuda", generator=gen)
    S = torch.triu(S, 1)
    S = S + S.t()
    S.fill_diagonal_(1.0)
    return S


def measured_fill(S, alpha):
    return float((S >= alpha).float().mean().item())


def main():
    n = int(os.environ.get("N", 50_000)); nt = int(os.environ.get("NT", 10_000)); folds = int(os.environ.get("FOLDS", 4096))
    alphas = [float(a) for a in os.environ.get("ALPHAS", "0.1,0.5,0.9").split(",")]
    f64 = os.environ.get("DTYPE", "f32") == "f64"      # the fp64 matrix instruction instead of the bf16 planes
    ss.init(0); ss.use_torch_stream()
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 4)
    S = sym_uniform(n, gen)
    if f64:
        S = S.double()
    yp, yi = rand_csr(n, nt, 0.01, gen)
    out = torch.empty((folds, nt), dtype=torch.float64 if f64 else torch.float32, device="cuda")
    for alpha in alphas:
        for weighted in (True, False):
            g = ss.DeviceGraph.from_similarity(None, S, (yp, yi, None, nt), alpha=alpha, weighted=weighted,
                                               dtype=np.float64 if f64 else np.float32)
            res = None
            for it in range(2):
                g.predict_loo(0, folds, clean=True, out=out)
                torch.cuda.synchronize()
                res = ss.timing_last()
            tf = 2.0 * folds * n * n / (res["transfer_ms"] * 1e-3) / 1e12
            print(json.dumps(dict(n=n, nt=nt, folds=folds, alpha=alpha, weighted=weighted, fill_measured=measured_fill(S, alpha), path=ss.path_last(),
                                  transfer_ms=res["transfer_ms"], spmm_ms=res["spmm_ms"], stage1_TFLOPs=tf)))
            g.close()
    if os.environ.get("CHECK", "1") == "1" and n <= 20000:
        from oracle import simspread_oracle as O
        import scipy.sparse as sp
        Sh = S.cpu().numpy()
        X = O.cutoff(Sh.astype(np.float64), float(alphas[-1]) if f64 else float(np.float32(alphas[-1])), False)   # the cutoff in the graph's precision
        Y = sp.csr_matrix((np.ones(yi.numel()), yi.cpu().numpy(), yp.cpu().numpy()), shape=(n, nt))
        qs = [0, folds // 2, folds - 1]
        want = O.predict_loo_dense(X, Y, clean_flag=True, queries=qs)
        got = out[qs].cpu().numpy()
        print("max rel err", np.abs(got - want).max() / np.abs(want).max())
if __name__ == "__main__":
    main()
