"""BASELINE.json configs 3-5 at (or near) their full sizes, through the same C ABI: a block of folds is scored
on the GPU and sampled folds are checked against the CPU oracle; size-independent properties cover the rest."""
import numpy as np
import pytest
import scipy.sparse as sp

import simspread_jl_amd as ss
from oracle import simspread_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    ss.init(0)
    ss.use_torch_stream()


def _host_csr(ptr, idx, val, shape):
    v = np.ones(idx.numel()) if val is None else val.cpu().numpy().astype(np.float64)
    return sp.csr_matrix((v, idx.cpu().numpy(), ptr.cpu().numpy()), shape=shape)


def test_config3_100k_loo_block_vs_oracle():
    """C3: 100k x 100k, 1 % (nnz 1e8 each), leave-one-out; one 512-fold block of the 100k folds."""
    import torch
    from tools.c3_loo import rand_csr, rand_sym_csr
    n, folds = 100_000, 512
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 3)
    xp, xi = rand_sym_csr(n, 0.01, gen)
    yp, yi = rand_csr(n, n, 0.01, gen)
    xv = (0.5 + 0.5 * torch.rand(xi.numel(), device="cuda", generator=gen)).float()
    g = ss.DeviceGraph.from_device_csr(0, n, n, n, None, (xp, xi, xv), (yp, yi, None), dtype=np.float32)
    assert g.nnz_xs > 9.9e7 and g.nnz_ys > 9.9e7
    lo = 50_000                                     # a block in the middle: what rank 4 of 8 would start with
    out = torch.empty((folds, n), dtype=torch.float32, device="cuda")
    g.predict_loo(lo, lo + folds, clean=True, out=out)
    X, Y = _host_csr(xp, xi, xv, (n, n)), _host_csr(yp, yi, None, (n, n))
    qs = [lo, lo + 255, lo + folds - 1]
    want = O.predict_loo_factored(X, Y, clean_flag=True, queries=qs)
    got = out[[q - lo for q in qs]].cpu().numpy().astype(np.float64)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-5
    assert ((want == 0) <= (got == 0)).all()
    # the same folds scored as part of a different block must give bit-identical rows (fold independence,
    # the property the 8-GPU sharding relies on)
    out2 = torch.empty((256, n), dtype=torch.float32, device="cuda")
    g.predict_loo(lo + 128, lo + 384, clean=True, out=out2)
    assert torch.equal(out2, out[128:384])


def test_config5_power_law_block_vs_oracle():
    """C5 at its specified weight: Zipf(1.2) source degrees (mean 1000, capped at Nt) and target popularity,
    nnz(Y) ~ 1e8 AFTER de-duplication, 200k nodes, hot rows/columns at random positions."""
    import torch
    from tools.c3_loo import rand_sym_csr
    from tools.c5_powerlaw import zipf_bipartite_spec
    n, folds = 100_000, 256
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 5)
    xp, xi = rand_sym_csr(n, 0.01, gen)
    yp, yi = zipf_bipartite_spec(n, n, 1000, 1.2, gen)
    assert 0.97e8 < yi.numel() < 1.03e8, yi.numel()        # the spec's weight, not a tenth of it
    xv = (0.5 + 0.5 * torch.rand(xi.numel(), device="cuda", generator=gen)).float()
    g = ss.DeviceGraph.from_device_csr(0, n, n, n, None, (xp, xi, xv), (yp, yi, None), dtype=np.float32)
    kf, ks, kt = g.degrees()
    rowdeg = (yp[1:] - yp[:-1]).cpu().numpy()
    assert kt.max() > 50_000 and rowdeg.max() > 90_000          # hot targets and (capped) hot sources
    assert np.median(rowdeg) < 400 and np.median(kt) < 1500     # ... and a long cold tail on both sides
    out = torch.empty((folds, n), dtype=torch.float32, device="cuda")
    g.predict_loo(0, folds, clean=True, out=out)
    assert "spmm_sell_sorted" in ss.path_last()                 # the skew-sorted, row-split stage-2 operand was chosen
    X, Y = _host_csr(xp, xi, xv, (n, n)), _host_csr(yp, yi, None, (n, n))
    hot = int(np.argmax(rowdeg[:folds]))                        # the heaviest source inside the block is a query too
    qs = sorted({0, 100, hot, folds - 1})
    want = O.predict_loo_factored(X, Y, clean_flag=True, queries=qs)
    got = out[qs].cpu().numpy().astype(np.float64)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-5
    assert ((want == -99) == (got == -99)).all()


def test_config4_dense_similarity_cutoff_sweep():
    """C4, small shape (12k sources, 512 folds -> the 128 x 128 bf16 kernel): raw similarity dense, cutoff sweep,
    MFMA stage 1, both weightings; S as SURVEY.md 8d states it (fill(S >= alpha) = 1 - alpha, measured)."""
    import torch
    from tools.c3_loo import rand_csr
    from tools.c4_dense import measured_fill, sym_uniform
    n, nt, folds = 12_000, 3_000, 512
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 4)
    S = sym_uniform(n, gen)
    yp, yi = rand_csr(n, nt, 0.01, gen)
    Sh = S.cpu().numpy().astype(np.float64)
    Y = _host_csr(yp, yi, None, (n, nt))
    out = torch.empty((folds, nt), dtype=torch.float32, device="cuda")
    for alpha, weighted in ((0.1, True), (0.5, False), (0.9, True)):
        assert abs(measured_fill(S, alpha) - (1.0 - alpha)) < 2e-3
        g = ss.DeviceGraph.from_similarity(None, S, (yp, yi, None, nt), alpha=alpha, weighted=weighted)
        g.predict_loo(1000, 1000 + folds, clean=True, out=out)
        assert "transfer_dense_bf16_128" in ss.path_last()
        X = O.cutoff(Sh, float(np.float32(alpha)), weighted)
        qs = [1000, 1255, 1000 + folds - 1]
        want = O.predict_loo_dense(X, Y, clean_flag=True, queries=qs)
        got = out[[q - 1000 for q in qs]].cpu().numpy().astype(np.float64)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (alpha, weighted)
        g.close()


@pytest.mark.parametrize("n", [20_000, 50_000])
def test_config4_ring_kernel_auto_selected(n):
    """C4 at production shapes: 4096 folds per alpha make (Mp/256) x (Np/256) >= 256 tiles, so the 256 x 256 ring
    kernel is chosen by size (NOT forced) and runs hundreds of K-tiles (20k: 313, 50k -- BASELINE configs[3] -- 782).
    90 % fill (alpha = 0.1, the named regime), both weightings, plus a sparse end of the sweep; sampled folds against
    the fp64 oracle (dense LOO form, three mat-vecs)."""
    import torch
    from tools.c3_loo import rand_csr
    from tools.c4_dense import measured_fill, sym_uniform
    nt, folds, lo = 10_000, 4096, 3000
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 4)
    S = sym_uniform(n, gen)
    yp, yi = rand_csr(n, nt, 0.01, gen)
    Sh = S.cpu().numpy()
    Y = _host_csr(yp, yi, None, (n, nt))
    out = torch.empty((folds, nt), dtype=torch.float32, device="cuda")
    cases = ((0.1, True), (0.1, False), (0.9, True)) if n <= 20_000 else ((0.1, True), (0.1, False))
    for alpha, weighted in cases:
        assert abs(measured_fill(S, alpha) - (1.0 - alpha)) < 2e-3
        g = ss.DeviceGraph.from_similarity(None, S, (yp, yi, None, nt), alpha=alpha, weighted=weighted)
        g.predict_loo(lo, lo + folds, clean=True, out=out)
        assert "transfer_dense_bf16_ring" in ss.path_last(), ss.path_last()
        X = O.cutoff(Sh, np.float32(alpha), weighted).astype(np.float64)   # thresholded in fp32 like the device, summed in fp64
        qs = [lo, lo + 2047, lo + folds - 1]
        want = O.predict_loo_dense(X, Y, clean_flag=True, queries=qs)
        del X
        got = out[[q - lo for q in qs]].cpu().numpy().astype(np.float64)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-5, (n, alpha, weighted)
        assert ((want == -99) == (got == -99)).all()
        g.close()


def test_config4_fp64_dense_path_at_20k():
    """The fp64 dense-similarity kernel (dense_f64.hip, v_mfma_f64_16x16x4_f64) at the shape profiles/ quotes it on:
    20k sources x 4096 folds = 1250 K-steps of 16 per tile, routed by the constructor (not forced), both weightings at
    the 90 % fill of the named regime plus the sparse end; sampled folds against the fp64 oracle to 1e-12."""
    import torch
    from tools.c3_loo import rand_csr
    from tools.c4_dense import measured_fill, sym_uniform
    n, nt, folds, lo = 20_000, 10_000, 4096, 3000
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 4)
    S = sym_uniform(n, gen).double()
    yp, yi = rand_csr(n, nt, 0.01, gen)
    Sh = S.cpu().numpy()
    Y = _host_csr(yp, yi, None, (n, nt))
    out = torch.empty((folds, nt), dtype=torch.float64, device="cuda")
    for alpha, weighted in ((0.1, True), (0.1, False), (0.9, True)):
        assert abs(measured_fill(S, alpha) - (1.0 - alpha)) < 2e-3
        g = ss.DeviceGraph.from_similarity(None, S, (yp, yi, None, nt), alpha=alpha, weighted=weighted, dtype=np.float64)
        g.predict_loo(lo, lo + folds, clean=True, out=out)
        assert "transfer_dense_f64_mfma" in ss.path_last(), ss.path_last()
        X = O.cutoff(Sh, alpha, weighted)
        qs = [lo, lo + 2047, lo + folds - 1]
        want = O.predict_loo_dense(X, Y, clean_flag=True, queries=qs)
        del X
        got = out[[q - lo for q in qs]].cpu().numpy()
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-12, (alpha, weighted)
        assert ((want == -99) == (got == -99)).all()
        g.close()
