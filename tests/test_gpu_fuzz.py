"""Randomised parity: shapes, densities and chunkings the hand-picked cases do not cover -- predict (query and source
rows), leave-one-out, every width route of the raw W*R SpMM, and the top-L reduction -- against the CPU oracle /
scipy / a stable sort.  Seeds are fixed: a failure names its case."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import simspread_jl_amd as ss
from oracle import simspread_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    ss.init(0)


def _close(got, want, dtype):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300) if want.size else 0.0
    assert err <= (1e-5 if dtype == np.float32 else 1e-12), err
    assert ((want == 0) <= (got == 0)).all()


@pytest.mark.parametrize("seed", range(12))
def test_predict_and_loo_random_graphs(seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    dtype = np.float32 if seed % 2 == 0 else np.float64
    nq, ns, nt = int(rng.integers(1, 300)), int(rng.integers(2, 1500)), int(rng.integers(1, 900))
    nf = ns if seed % 3 else int(rng.integers(2, 1200))
    dx, dy = float(rng.choice([0.01, 0.05, 0.2, 0.6])), float(rng.choice([0.005, 0.03, 0.3]))
    weighted = bool(rng.integers(0, 2))
    if seed % 4 == 1:   # several LDS chunks and transfer batches
        monkeypatch.setenv("SS_SELL_CHUNK", str(int(rng.choice([64, 200, 900]))))
        monkeypatch.setenv("SS_TRANSFER_CHUNK", str(int(rng.choice([32, 100, 400]))))
        monkeypatch.setenv("SS_TRANSFER_BYTES", str(1 << 18))
    Xq, Xs, Ys = O.synth_bipartite(nq, ns, nf, nt, dx, dy, seed=seed, weighted=weighted, dtype=dtype)
    f64 = [m.astype(np.float64) for m in (Xq, Xs, Ys)]
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=dtype)
    _close(g.predict("query"), O.predict_factored(*f64, rows="query"), dtype)
    want = O.predict_factored(*f64, rows="source")
    kt = O.degrees(f64[1], f64[2])[2]
    want[:, kt == 0] = -99.0
    _close(g.predict("source", clean=True), want, dtype)
    if nf == ns:   # leave-one-out on the square similarity
        X = f64[1].tolil(); X.setdiag(1.0); X = X.tocsr()
        gl = ss.DeviceGraph.from_sparse(None, X.astype(dtype), Ys, dtype=dtype)
        lo = int(rng.integers(0, ns)); hi = int(min(ns, lo + rng.integers(1, 200)))
        got = gl.predict_loo(lo, hi, clean=True)
        _close(got, O.predict_loo_factored(X, f64[2], clean_flag=True, queries=range(lo, hi)), dtype)


@pytest.mark.parametrize("seed", range(10))
def test_spmm_random_shapes_every_route(seed, monkeypatch):
    rng = np.random.default_rng(2000 + seed)
    dtype = np.float32 if seed % 2 == 0 else np.float64
    M, K = int(rng.integers(1, 5000)), int(rng.integers(1, 9000))
    W = sp.random(M, K, density=float(rng.choice([0.0005, 0.004, 0.03, 0.2])), format="csr", random_state=rng, dtype=np.float64)
    W.data = np.ones(W.nnz) if seed % 3 == 0 else rng.random(W.nnz) + 0.5
    if M > 3:
        W = W.tolil(); W[int(rng.integers(0, M)), :] = 1.0; W[int(rng.integers(0, M)), :] = 0.0; W = W.tocsr()
    if seed % 2:
        monkeypatch.setenv("SS_NARROW_CHUNK", str(int(rng.choice([64, 100, 333, 1000]))))
        monkeypatch.setenv("SS_SELL_CHUNK", str(int(rng.choice([64, 500, 2000]))))
    w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
    for B in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 32, 33, 64, 65, 130):
        R = rng.standard_normal((K, B))
        got = np.asarray(w.spmm(R.astype(dtype)), np.float64)
        scale = max((abs(W) @ np.abs(R)).max(), 1e-300)
        assert np.abs(got - W @ R).max() / scale <= (2e-6 if dtype == np.float32 else 1e-13), (B, ss.path_last())


@pytest.mark.parametrize("seed", range(8))
def test_topl_random_rows(seed):
    rng = np.random.default_rng(3000 + seed)
    rows, n = int(rng.integers(1, 40)), int(rng.integers(1, 120000))
    x = rng.standard_normal((rows, n)).astype(np.float32)
    if seed % 4 == 1: x = np.round(x * 3) / 3
    if seed % 4 == 2: x[:, rng.random(n) < 0.9] = 0.0
    if seed % 4 == 3: x[int(rng.integers(0, rows))] = -99.0
    for L in {1, min(n, 100), min(n, 1024)}:
        idx, val = ss.topl(x, L)
        want = np.argsort(-x, axis=1, kind="stable")[:, :L]
        np.testing.assert_array_equal(idx, want)
        np.testing.assert_array_equal(val, np.take_along_axis(x, want, 1))
