"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Tolerances: fp64 results must
agree to 1e-12 relative; fp32 results to 1e-5 relative to the largest score of the block (north_star:
1e-5 relative fp64 tolerance), exact zeros must stay exact zeros."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

import simspread_jl_amd as ss
from oracle import simspread_oracle as O
from oracle import c_oracle

pytestmark = pytest.mark.gpu

TOL = {np.float32: 1e-5, np.float64: 1e-12}


def assert_close(got, want, dtype):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    scale = max(np.abs(want).max(), 1e-300)
    err = np.abs(got - want).max() / scale
    assert err <= TOL[dtype], f"max err {err:.3e} (relative to the largest score)"
    assert ((want == 0) <= (got == 0)).all(), "a structurally zero score became non-zero"


@pytest.fixture(scope="module", autouse=True)
def _init():
    ss.init(0)


# ----------------------------------------------------------------------------- reference KATs through the mirror
def test_kats_through_the_host_mirror(kats):
    M = np.array(kats["k"]["M"], float)
    assert int(ss.k(1, M)) == 0 and int(ss.k(M[0])) == 0
    assert ss.k(M).ravel().tolist() == kats["k"]["row_degrees"]
    c = kats["cutoff"]
    y, z = np.array(c["y"]).reshape(-1, 1), np.array(c["z"])
    for case in c["cases"]:
        a = case["alpha"]
        assert ss.cutoff(c["x"], a, False) == pytest.approx(case["x_bin"])
        assert ss.cutoff(c["x"], a, True) == pytest.approx(case["x_w"])
        np.testing.assert_array_equal(ss.cutoff(y, a, False).ravel(), case["y_bin"])
        np.testing.assert_array_equal(ss.cutoff(y, a, True).ravel(), case["y_w"])
        np.testing.assert_array_equal(ss.cutoff(z, a, False), case["z_bin"])
        np.testing.assert_array_equal(ss.cutoff(z, a, True), case["z_w"])
    f = kats["featurize"]
    M0 = ss.NamedMatrix(f["M0"], f["rows"], f["cols"])
    assert ss.featurize(M0, f["alpha"], False) == ss.NamedMatrix(f["unweighted"], f["rows"], f["out_cols"])
    assert ss.featurize(M0, f["alpha"], True) == ss.NamedMatrix(f["weighted"], f["rows"], f["out_cols"])
    assert ss.featurize(M0, f["alpha"]).names(2) == f["out_cols"]
    s = kats["spread"]
    got = ss.spread(np.array(s["M"], float))
    assert np.linalg.norm(got - np.array(s["W"])) <= s["rtol"] * np.linalg.norm(got)


def test_construct_kat(kats):
    c = kats["construct"]
    X = ss.NamedMatrix(c["X"], c["X_rows"], c["X_cols"])
    y = ss.NamedMatrix(c["y"], c["y_rows"], c["y_cols"])
    A, B = ss.construct(y, X, c["queries"])
    assert A.names(1) == A.names(2) == c["node_order"]
    assert B.names(1) == B.names(2) == c["node_order"]
    X2 = ss.NamedMatrix(c["X"], c["X_rows"], c["X_rows"])
    with pytest.raises(AssertionError, match=c["same_names_message"]):
        ss.construct(y, X2, c["queries"])
    X3 = ss.NamedMatrix(c["row_mismatch_X"], c["row_mismatch_rows"], c["X_cols"])
    with pytest.raises(AssertionError, match=c["row_mismatch_message"]):
        ss.construct(y, X3, c["queries"])
    # dense view equals the oracle's literal block matrices
    Ao, Bo = O.construct_queries(O.Named(c["y"], c["y_rows"], c["y_cols"]), O.Named(c["X"], c["X_rows"], c["X_cols"]),
                                 c["queries"])
    np.testing.assert_array_equal(A.array, Ao.array)
    np.testing.assert_array_equal(B.array, Bo.array)


def test_predict_kat_exact_both_call_forms(kats):
    p = kats["predict"]
    A = ss.NamedMatrix(p["A"], p["names"], p["names"])
    B = ss.NamedMatrix(p["B"], p["names"], p["names"])
    y = ss.NamedMatrix(p["y"], p["y_rows"], p["y_cols"])
    want = ss.NamedMatrix(p["yhat"], p["y_rows"], p["y_cols"])
    assert ss.predict(A, B, y) == want           # exact ==, like test/runtests.jl:156
    assert ss.predict((A, B), y) == want
    assert ss.predict(A, B, y, GPU=True) == want  # fp32 path: values are exactly representable


def test_clean_kat(kats):
    c = kats["clean"]
    A = ss.NamedMatrix(c["A"], c["names"], c["names"])
    y = ss.NamedMatrix(c["y"], c["y_rows"], c["y_cols"])
    yhat = ss.NamedMatrix(c["yhat_in"], c["y_rows"], c["y_cols"])
    ss.clean(yhat, A, y)
    assert yhat == ss.NamedMatrix(c["yhat_out"], c["y_rows"], c["y_cols"])


# ----------------------------------------------------------------------------- tutorial-shaped chain vs the literal oracle
@pytest.mark.parametrize("weighted", [False, True])
def test_featurize_construct_predict_chain_vs_literal(weighted):
    rng = np.random.default_rng(11)
    n, nt, ntest = 40, 9, 7
    S = rng.random((n, n)); S = (S + S.T) / 2; np.fill_diagonal(S, 1.0)
    nm = [f"D{i:03d}" for i in range(n)]
    train, test = nm[:-ntest], nm[-ntest:]
    Y = (rng.random((n, nt)) < 0.3).astype(float)
    tn = [f"T{i}" for i in range(nt)]
    # host mirror
    DD, yy = ss.NamedMatrix(S, nm, nm), ss.NamedMatrix(Y, nm, tn)
    Xtr = ss.featurize(DD.sub(train, train), 0.55, weighted)
    Xte = ss.featurize(DD.sub(test, train), 0.55, weighted)
    G = ss.construct(yy.sub(train, tn), yy.sub(test, tn), Xtr, Xte)
    got_te = ss.predict(G, yy.sub(test, tn))
    got_tr = ss.predict(G, yy.sub(train, tn))
    # literal oracle
    oD, oy = O.Named(S, nm, nm), O.Named(Y, nm, tn)
    oXtr = O.featurize(oD.sub(train, train), 0.55, weighted)
    oXte = O.featurize(oD.sub(test, train), 0.55, weighted)
    A, B = O.construct_split(oy.sub(train, tn), oy.sub(test, tn), oXtr, oXte)
    assert_close(got_te.array, O.predict(A, B, oy.sub(test, tn)).array, np.float64)
    assert_close(got_tr.array, O.predict(A, B, oy.sub(train, tn)).array, np.float64)
    assert got_te.names(1) == test and got_te.names(2) == tn
    # 3-layer graph, predict(A, ytrain)
    A3 = ss.construct(yy.sub(train, tn), Xtr)
    got3 = ss.predict(A3, yy.sub(train, tn))
    assert_close(got3.array, O.predict_single(O.construct_single(oy.sub(train, tn), oXtr), oy.sub(train, tn)).array,
                 np.float64)
    # mixed query + source rows in one call, permuted columns
    rows = [test[2], train[5], test[0]]
    cols = tn[::-1]
    got_mix = ss.predict(G, yy.sub(rows, cols))
    assert_close(got_mix.array, O.predict(A, B, oy.sub(rows, cols)).array, np.float64)


def test_predict_general_path_and_clean_outside_the_targets(kats):
    """The branches julia/SimSpreadDevice.jl takes when the block asked for is NOT [queries|sources] x targets -- the
    same ones core.py takes, pinned here against the literal oracle so that a maintainer can diff behaviours (no Julia
    runs here): (1) predict((A,B), y) with feature ROWS or source COLUMNS falls back to the general A*(W*W) path
    (_node_groups cannot place the names); (2) clean!(yhat, A, y) with column names that are not targets uses the
    degree of those rows of the dense A (src/core.jl:479 looks at A[name, :] whatever the name is)."""
    c = kats["construct"]
    X, y = ss.NamedMatrix(c["X"], c["X_rows"], c["X_cols"]), ss.NamedMatrix(c["y"], c["y_rows"], c["y_cols"])
    oX, oy = O.Named(c["X"], c["X_rows"], c["X_cols"]), O.Named(c["y"], c["y_rows"], c["y_cols"])
    A, B = ss.construct(y, X, c["queries"])
    oA, oB = O.construct_queries(oy, oX, c["queries"])
    nodes = c["node_order"]
    feats = [n for n in nodes if n in set(X.cols) and n in set(A.names(1))]
    srcs = [n for n in y.rows if n not in c["queries"]]
    # (1) rows = features, columns = sources: nothing of it is a query/source x target block
    blk = ss.NamedMatrix(np.zeros((len(feats), len(srcs))), feats, srcs)
    oblk = O.Named(np.zeros((len(feats), len(srcs))), feats, srcs)
    got = ss.predict((A, B), blk)
    want = O.predict(oA, oB, oblk)
    assert got.names(1) == feats and got.names(2) == srcs
    np.testing.assert_allclose(got.array, want.array, rtol=1e-12, atol=1e-15)
    # mixed: one query row, one feature row (the feature row forces the general path for the whole call)
    rows = [c["queries"][0], feats[0]]
    blk2 = ss.NamedMatrix(np.zeros((2, len(y.cols))), rows, list(y.cols))
    got2 = ss.predict(A, B, blk2)
    want2 = O.predict(oA, oB, O.Named(np.zeros((2, len(y.cols))), rows, list(y.cols)))
    np.testing.assert_allclose(got2.array, want2.array, rtol=1e-12, atol=1e-15)
    # (2) clean! with non-target column names: the degree of those rows of A decides
    yh = ss.NamedMatrix(np.ones((1, len(feats))), [c["queries"][0]], feats)
    oyh = O.Named(np.ones((1, len(feats))), [c["queries"][0]], feats)
    ss.clean(yh, A, yh)
    O.clean(oyh, oA, oyh)
    np.testing.assert_array_equal(yh.array, oyh.array)


# ----------------------------------------------------------------------------- engine parity on seeded synthetic graphs
CASES = [
    # nq, ns, nf, nt, dx, dy, weighted
    (33, 70, 70, 41, 0.15, 0.10, True),
    (128, 257, 200, 130, 0.08, 0.04, False),
    (5, 64, 64, 1, 0.3, 0.5, True),
    (1, 3, 3, 2, 0.9, 0.9, False),
    (200, 1000, 1000, 777, 0.05, 0.01, True),
]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", CASES)
def test_predict_query_and_source_rows(case, dtype):
    nq, ns, nf, nt, dx, dy, weighted = case
    Xq, Xs, Ys = O.synth_bipartite(nq, ns, nf, nt, dx, dy, seed=100 + nq, weighted=weighted, dtype=dtype)
    # make a zero-degree feature, an isolated source and an empty target when the graph is big enough
    if ns > 10:
        Xs = Xs.tolil(); Ys = Ys.tolil()
        Xs[:, 1] = 0; Xs[2, :] = 0; Ys[2, :] = 0; Ys[:, nt - 1] = 0
        Xs = Xs.tocsr(); Ys = Ys.tocsr()
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=dtype)
    f64 = [m.astype(np.float64) for m in (Xq, Xs, Ys)]
    assert_close(g.predict("query"), O.predict_factored(*f64, rows="query"), dtype)
    assert_close(g.predict("source"), O.predict_factored(*f64, rows="source"), dtype)
    kf, ks, kt = g.degrees()
    okf, oks, okt = O.degrees(f64[1], f64[2])
    np.testing.assert_array_equal(kf, okf); np.testing.assert_array_equal(ks, oks); np.testing.assert_array_equal(kt, okt)
    # sub-range + fused clean! + column-major (Julia) layout
    lo, hi = nq // 3, nq
    want = O.predict_factored(*f64, rows="query")[lo:hi].copy()
    want[:, okt == 0] = -99.0
    got = g.predict("query", lo, hi, clean=True, layout="col")
    assert got.flags.f_contiguous or got.shape[0] == 1 or got.shape[1] == 1
    assert_close(got, want, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dense_assembly_with_fused_cutoff_matches_featurize(dtype):
    rng = np.random.default_rng(5)
    nq, ns, nt = 37, 150, 60
    Sq, Ss = rng.random((nq, ns)).astype(dtype), rng.random((ns, ns)).astype(dtype)
    Y = (rng.random((ns, nt)) < 0.05).astype(dtype)
    Ss[4, 9] = 0.7; Sq[0, 0] = 0.7   # exactly alpha: `>=` keeps it
    for weighted in (False, True):
        g = ss.DeviceGraph.from_dense(Sq, Ss, Y, alpha=dtype(0.7), weighted=weighted, dtype=dtype)
        Xq = O.cutoff(Sq.astype(np.float64), float(dtype(0.7)), weighted)
        Xs = O.cutoff(Ss.astype(np.float64), float(dtype(0.7)), weighted)
        assert g.nnz_xq == np.count_nonzero(Xq) and g.nnz_xs == np.count_nonzero(Xs)
        assert_close(g.predict("query"), O.predict_factored(Xq, Xs, Y.astype(np.float64)), dtype)


def test_results_are_bitwise_reproducible():
    # no float atomics anywhere: every sum has a fixed order, so repeated runs and fresh handles agree bit for bit
    Xq, Xs, Ys = O.synth_bipartite(300, 2000, 2000, 900, 0.05, 0.02, seed=31, dtype=np.float32)
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    a = g.predict("query").copy()
    for _ in range(3):
        assert np.array_equal(g.predict("query"), a)
    g2 = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    assert np.array_equal(g2.predict("query"), a)
    s = g.predict("source").copy()
    assert np.array_equal(g2.predict("source"), s)


def test_stage1_bank_schedule_changes_no_bit(monkeypatch):
    """The stage-1 operand orders the entries of every sub-row by LDS bank (chunk_fill_kernel); every accumulator still
    receives the same addends in the same order, so scores must be bit-identical with and without it."""
    Xq, Xs, Ys = O.synth_bipartite(257, 3000, 3000, 700, 0.05, 0.02, seed=33, dtype=np.float32)
    monkeypatch.setenv("SS_CHUNK_SCHED", "0")
    g0 = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    a, sa = g0.predict("query").copy(), g0.predict("source").copy()
    monkeypatch.setenv("SS_CHUNK_SCHED", "1")
    g1 = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    assert np.array_equal(g1.predict("query"), a)
    assert np.array_equal(g1.predict("source"), sa)
    monkeypatch.setenv("SS_TRANSFER_CHUNK", "96")      # sub-rows of a few entries: mostly below the 32-entry threshold
    g2 = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    want = O.predict_factored(Xq.astype(np.float64), Xs.astype(np.float64), Ys.astype(np.float64), "query")
    assert np.abs(g2.predict("query") - want).max() / np.abs(want).max() < 1e-5


def test_two_threads_two_handles():
    """Handle-scoped locking: two host threads, each predicting on its own handle at the same time (their kernels
    interleave on the library stream), must produce exactly what the same calls give one after the other."""
    import threading
    graphs, want = [], []
    for seed in (41, 42):
        Xq, Xs, Ys = O.synth_bipartite(400, 1500, 1500, 600, 0.05, 0.02, seed=seed, dtype=np.float32)
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
        graphs.append(g)
        want.append((g.predict("query").copy(), g.predict("source").copy()))
    errors = []

    def worker(i):
        try:
            for _ in range(20):
                q = graphs[i].predict("query")
                s = graphs[i].predict("source")
                if not (np.array_equal(q, want[i][0]) and np.array_equal(s, want[i][1])):
                    errors.append((i, "mismatch"))
        except Exception as e:
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert not errors, errors[:3]
    # the same handle from two threads: calls queue up on the handle's lock, results still exact
    threads = [threading.Thread(target=worker, args=(0,)) for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors[:3]


def test_in_library_rccl_gather_single_rank():
    """ss_comm_* / ss_gather_rows_*: the library's own RCCL communicator (dlopen'ed).  A one-GPU box allows exactly one
    rank per device, so this covers loading RCCL, creating / destroying the communicator, the own-block path of the
    exchange and the argument checks; the peer-to-peer legs run in the driver's multi-GPU job only."""
    import torch
    ss.use_torch_stream()
    with pytest.raises(ss.SimSpreadError):
        ss.lib_gather_scores(torch.zeros((2, 3), device="cuda"), [2])          # no communicator yet
    uid = ss.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ss.comm_init(uid, 0, 1)
    for dt in (torch.float32, torch.float64):
        local = torch.arange(5 * 7, device="cuda", dtype=dt).reshape(5, 7).contiguous()
        assert torch.equal(ss.lib_gather_scores(local, [5]), local)             # every rank receives
        assert torch.equal(ss.lib_gather_scores(local, [5], root=0), local)     # gather to rank 0
    with pytest.raises(ValueError):
        ss.lib_gather_scores(local, [4])
    ss.comm_destroy()
    with pytest.raises(ss.SimSpreadError):
        ss.lib_gather_scores(local, [5])
    with pytest.raises(ss.SimSpreadError):
        ss.comm_init(uid, 3, 2)                                                  # rank outside the communicator


def test_multi_chunk_and_row_batches(monkeypatch):
    # force several LDS chunks of W's columns and several transfer batches; results must not change
    Xq, Xs, Ys = O.synth_bipartite(150, 700, 700, 300, 0.05, 0.03, seed=9, dtype=np.float32)
    want = O.predict_factored(*(m.astype(np.float64) for m in (Xq, Xs, Ys)))
    monkeypatch.setenv("SS_SELL_CHUNK", "128")
    monkeypatch.setenv("SS_TRANSFER_BYTES", str(1 << 20))
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    got = g.predict("query", clean=True)
    kt = O.degrees(Xs, Ys)[2]
    want[:, kt == 0] = -99.0
    assert_close(got, want, np.float32)


def test_wide_source_row_needs_chunked_transfer():
    # ns larger than one LDS accumulator of the transfer kernel (16384 floats)
    ns, nq, nt = 40000, 16, 50
    Xq, Xs, Ys = O.synth_bipartite(nq, ns, 300, nt, 0.02, 0.001, seed=21, dtype=np.float32)
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    assert_close(g.predict("query"), O.predict_factored(*(m.astype(np.float64) for m in (Xq, Xs, Ys))), np.float32)


def test_explicit_zeros_are_not_edges():
    Xq, Xs, Ys = O.synth_bipartite(10, 40, 40, 12, 0.2, 0.2, seed=2, dtype=np.float64)
    Xs2 = Xs.copy(); Xs2.data[::3] = 0.0     # stored zeros: not neighbours, not counted in degrees
    g = ss.DeviceGraph.from_sparse(Xq, Xs2, Ys, dtype=np.float64)
    Xs3 = Xs2.copy(); Xs3.eliminate_zeros()
    assert g.nnz_xs == Xs3.nnz
    assert_close(g.predict("query"), O.predict_factored(Xq, Xs3, Ys), np.float64)


def test_empty_and_degenerate_inputs():
    e = sp.csr_matrix((0, 5)); Xs = sp.csr_matrix((4, 5)); Ys = sp.csr_matrix((4, 3))
    g = ss.DeviceGraph.from_sparse(e, Xs, Ys, dtype=np.float32)
    assert g.predict("query").shape == (0, 3)
    np.testing.assert_array_equal(g.predict("source"), np.zeros((4, 3), np.float32))
    np.testing.assert_array_equal(g.predict("source", clean=True), np.full((4, 3), -99, np.float32))


def test_bad_inputs_are_rejected_not_faulted():
    from simspread_jl_amd import _lib as L
    import ctypes as C
    lib = L.lib()
    ptr = np.array([0, 2, 3], np.int64); idx = np.array([3, 1, 0], np.int32); val = np.ones(3, np.float32)
    h = C.c_void_p()
    z = np.zeros(1, np.int64)
    def mk(p, i, cols):
        return lib.ss_spmat_create_csr_f32(2, cols, p.ctypes.data, i.ctypes.data, val.ctypes.data, 0, 0, C.byref(h))
    assert mk(ptr, idx, 4) == -1 and "increasing" in lib.ss_last_error().decode()
    assert mk(ptr, np.array([0, 9, 1], np.int32), 4) == -1 and "range" in lib.ss_last_error().decode()
    assert mk(np.array([0, 5, 3], np.int64), idx, 4) == -1
    g = ss.DeviceGraph.from_sparse(sp.csr_matrix((2, 3)), sp.csr_matrix((3, 3)), sp.csr_matrix((3, 2)))
    with pytest.raises(ss.SimSpreadError, match="row range"):
        g.predict("query", 0, 5)
    with pytest.raises(ss.SimSpreadError, match="leave-one-out"):
        g.predict_loo()
    g64 = ss.DeviceGraph.from_sparse(None, sp.identity(3, format="csr"), sp.csr_matrix((3, 2)), dtype=np.float64)
    out = np.zeros((3, 2), np.float32)
    assert lib.ss_predict_f32(g64._h, 1, 0, 3, 0, out.ctypes.data, 2, 0, 0) == -1  # precision mismatch


# ----------------------------------------------------------------------------- leave-one-out
def _iris():
    here = os.path.join(os.path.dirname(__file__), "golden", "iris")
    def read(p):
        with open(os.path.join(here, p)) as f:
            lines = f.read().splitlines()
        return [l.split()[0] for l in lines[1:]], lines[0].split(), np.array([[float(v) for v in l.split()[1:]] for l in lines[1:]])
    rows, _, F = read("iris.features")
    _, cc, Cm = read("iris.classes")
    S = np.minimum(F[:, None, :], F[None, :, :]).sum(-1) / np.maximum(F[:, None, :], F[None, :, :]).sum(-1)
    return rows, cc, S, Cm


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("weighted", [False, True])
def test_iris_full_loo_vs_per_fold_reference_semantics(dtype, weighted):
    rows, cc, S, Cm = _iris()
    g = ss.DeviceGraph.from_dense(None, S.astype(dtype), Cm.astype(dtype), alpha=dtype(0.9), weighted=weighted, dtype=dtype)
    got = g.predict_loo(clean=True)
    X = O.cutoff(S.astype(dtype).astype(np.float64), float(dtype(0.9)), weighted)
    want = O.predict_loo_factored(X, Cm, clean_flag=True)
    assert_close(got, want, dtype)
    # three folds the long way round: literal construct(y, X, [q]) + predict + clean!
    Xn, Yn = O.Named(X, rows, ["f" + r for r in rows]), O.Named(Cm, rows, cc)
    for i in (0, 77, 149):
        A, B = O.construct_queries(Yn, Xn, [rows[i]])
        yq = Yn.sub([rows[i]], cc)
        yh = O.predict(A, B, yq); O.clean(yh, A, yq)
        assert_close(got[i:i + 1], yh.array, dtype)


def test_loo_clean_flags_targets_whose_only_edge_is_the_query():
    rng = np.random.default_rng(3)
    n, nt = 90, 20
    S = rng.random((n, n)); S = (S + S.T) / 2; np.fill_diagonal(S, 1.0)
    Y = (rng.random((n, nt)) < 0.1).astype(np.float64)
    Y[:, 4] = 0; Y[17, 4] = 1      # single edge
    Y[:, 5] = 0                    # no edge at all
    X = O.cutoff(S, 0.6, True)
    g = ss.DeviceGraph.from_dense(None, S, Y, alpha=0.6, weighted=True, dtype=np.float64)
    got = g.predict_loo(10, 60, clean=True)
    want = O.predict_loo_factored(X, Y, clean_flag=True, queries=range(10, 60))
    assert_close(got, want, np.float64)
    assert got[7, 4] == -99 and (got[:, 5] == -99).all() and (got[np.arange(50) != 7, 4] != -99).all()


def test_loo_through_the_mirror_equals_fold_loop():
    rng = np.random.default_rng(8)
    n, nt = 30, 5
    S = rng.random((n, n)); S = (S + S.T) / 2; np.fill_diagonal(S, 1.0)
    nm = [f"d{i:02d}" for i in range(n)]; tn = [f"t{i}" for i in range(nt)]
    y = ss.NamedMatrix((rng.random((n, nt)) < 0.3).astype(float), nm, tn)
    X = ss.featurize(ss.NamedMatrix(S, nm, nm), 0.5, True)
    g = ss.DeviceGraph.from_dense(None, X.array, y.array, dtype=np.float64)
    fast = g.predict_loo(clean=True)
    for i in (0, 13, 29):
        A, B = ss.construct(y, X, [nm[i]])
        yq = y.sub([nm[i]], tn)
        yh = ss.predict((A, B), yq)
        ss.clean(yh, A, yq)
        assert_close(fast[i:i + 1], yh.array, np.float64)


def test_save_loo_streams_the_reference_fold_loop_output(tmp_path):
    """save_loo: every LOO fold scored from one resident graph, block by block, and appended in save's wire format
    (src/core.jl:542-561) -- the file must be what the reference's user loop writes fold by fold (construct ->
    predict -> clean! -> save), compared field by field (same text for names, fold ids and labels; scores to 1e-12)."""
    rng = np.random.default_rng(9)
    n, nt = 23, 4
    S = rng.random((n, n)); S = (S + S.T) / 2; np.fill_diagonal(S, 1.0)
    nm = [f"d{i:02d}" for i in range(n)]; tn = [f"t{i}" for i in range(nt)]
    Y = (rng.random((n, nt)) < 0.3).astype(np.int64)
    Y[:, 3] = 0; Y[4, 3] = 1                                  # a target whose only edge is one query: -99.0 in its fold
    y = ss.NamedMatrix(Y, nm, tn)                             # integer payload: labels print bare, like test/data/save1-4
    X = ss.featurize(ss.NamedMatrix(S, nm, nm), 0.5, True)
    fast, slow = tmp_path / "fast.tsv", tmp_path / "slow.tsv"
    assert ss.save_loo(str(fast), y, X, block=7) == n * nt    # 4 blocks, the last one short
    for i, s in enumerate(nm, start=1):
        A, B = ss.construct(y, X, [s])
        yq = y.sub([s], tn)
        yh = ss.predict((A, B), yq)
        ss.clean(yh, A, yq)
        ss.save(str(slow), i, yh, yq)
    a = [ln.split("\t") for ln in fast.read_text().splitlines()]
    b = [ln.split("\t") for ln in slow.read_text().splitlines()]
    assert len(a) == len(b) == n * nt
    for la, lb in zip(a, b):
        assert la[:3] == lb[:3] and la[4] == lb[4], (la, lb)
        assert abs(float(la[3]) - float(lb[3])) <= 1e-12 * max(1.0, abs(float(lb[3])))
    assert any(l[3] == "-99.0" for l in a) and a[0][0] == "1" and a[-1][0] == str(n)
    assert all("." in l[3] and "." not in l[4] for l in a)    # Float64 scores, bare Int labels


# ----------------------------------------------------------------------------- raw W*R SpMM
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("B", [1, 2, 3, 4, 7, 8, 12, 16, 32, 33, 64, 65, 200])
def test_spmm_against_scipy(B, dtype):
    rng = np.random.default_rng(B)
    M, K = 517, 389
    W = sp.random(M, K, density=0.06, format="csr", random_state=rng, dtype=np.float64)
    W.data = rng.random(W.nnz) + 0.5
    if B % 2 == 0:
        W.data[:] = 1.0  # pattern-only operand (binary Y)
    R = rng.standard_normal((K, B))
    w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
    want = W @ R
    assert_close_signed(w.spmm(R.astype(dtype)), want, dtype)                       # row-major operands
    assert_close_signed(w.spmm(np.ascontiguousarray(R.T).astype(dtype), colmajor=True).T, want, dtype)
    by, fl = w.cost(B)
    vb = np.dtype(dtype).itemsize
    assert by == W.nnz * (vb + 4) + (M + 1) * 4 + K * B * vb + M * B * vb and fl == 2.0 * W.nnz * B


def assert_close_signed(got, want, dtype):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    # signed operands: bound the error by the sum of magnitudes, not by the (possibly cancelling) result
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    assert err <= (2e-5 if dtype == np.float32 else 1e-12), err


def test_spmm_empty_rows_long_rows_and_multichunk(monkeypatch):
    rng = np.random.default_rng(0)
    M, K = 300, 5000
    rows = []
    for m in range(M):
        n = 0 if m % 7 == 0 else (K if m == 11 else int(rng.integers(1, 40)))
        cols = np.sort(rng.choice(K, n, replace=False))
        rows.append(cols)
    indptr = np.cumsum([0] + [len(r) for r in rows])
    W = sp.csr_matrix((rng.random(indptr[-1]) + 0.1, np.concatenate(rows), indptr), shape=(M, K))
    R = rng.random((K, 70))
    monkeypatch.setenv("SS_SELL_CHUNK", "1000")
    monkeypatch.setenv("SS_NARROW_CHUNK", "700")
    w = ss.DeviceSpMat(W, dtype=np.float64)
    assert_close(w.spmm(R), W @ R, np.float64)
    for B in (1, 2, 5, 8, 16):    # narrow kernel (B <= 4: 8 chunks, partial sums combined in order) / 2-D kernel
        assert_close(w.spmm(R[:, :B].copy()), W @ R[:, :B], np.float64)
    w32 = ss.DeviceSpMat(W, dtype=np.float32)
    for B in (1, 4, 8, 9, 17, 32, 40, 64):   # 5..64: 2-D kernel over 8 LDS chunks of R
        assert_close(w32.spmm(R[:, :B].astype(np.float32)), W @ R[:, :B], np.float32)
    # skewed rows: split the long rows into virtual rows, sort by length, put the scores back in order
    monkeypatch.setenv("SS_SELL_SORT", "1")
    monkeypatch.setenv("SS_SELL_LMAX", "256")
    for dt in (np.float64, np.float32):
        ws = ss.DeviceSpMat(W, dtype=dt)
        assert_close(ws.spmm(R.astype(dt)), W @ R, dt)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mean_entries", [6, 40, 100, 210, 380])
def test_spmm_narrow_every_group_size(mean_entries, dtype):
    """The HBM-bound narrow kernel picks 8 / 16 / 32 / 64 lanes per row (and two requests per lane) from the mean sub-row
    length; every variant runs its software pipeline (quads of the next row group in flight, bounds of the one after)
    over a row count that is no multiple of a step, with empty rows and a few rows far longer than the requests
    cover (rolled tail)."""
    rng = np.random.default_rng(mean_entries)
    M, K = 1237, 6000
    lens = rng.poisson(mean_entries, M)
    lens[::13] = 0
    lens[5] = min(K, 6 * mean_entries + 300)
    lens[M - 1] = min(K, 5 * mean_entries + 100)
    rows = [np.sort(rng.choice(K, int(n), replace=False)) for n in lens]
    indptr = np.cumsum([0] + [len(r) for r in rows])
    W = sp.csr_matrix((rng.random(indptr[-1]) + 0.1, np.concatenate(rows), indptr), shape=(M, K))
    R = rng.standard_normal((K, 4))
    w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
    for B in (1, 2, 3, 4):
        assert_close_signed(w.spmm(R[:, :B].astype(dtype)), W @ R[:, :B], dtype)
        # (B = 3, 4 went to the lane-per-row kernel with four columns per tile row in round 3; SS_CSELL_ROW16=0 keeps them here)
        assert ("spmm_csell" if (B >= 3 or (B == 2 and dtype == np.float64)) else "spmm_chunked_narrow") in ss.path_last()


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("dtype,B", [(np.float32, 12), (np.float32, 16), (np.float32, 28), (np.float32, 64), (np.float32, 52),
                                     (np.float64, 6), (np.float64, 16), (np.float64, 32)])
def test_spmm_colgroup_2d_cut(dtype, B, binary, monkeypatch):
    """Mid width (fp32 5 <= B <= 64, fp64 5 <= B <= 32): the 2-D kernel (spmm_colgroup.hip; fp32 reaches it under
    SS_CSELL=0 since round 3).  Several row blocks, several chunk groups (partial sums combined in fixed order), several
    LDS chunks of R per group, sub-rows longer than the prefetched batches (row 7 is full) and empty rows; must agree with
    scipy and with the SELL kernel, and be bitwise repeatable."""
    rng = np.random.default_rng(100 + B)
    M, K = 9011, 2900
    W = sp.random(M, K, density=0.012, format="lil", random_state=rng, dtype=np.float64)
    W[7, :] = 1.0
    W[100:130, :] = 0.0
    W = W.tocsr()
    W.data = np.ones(W.nnz) if binary else rng.random(W.nnz) + 0.5
    W.eliminate_zeros()
    R = rng.standard_normal((K, B))
    want = W @ R
    monkeypatch.setenv("SS_CSELL", "0")
    monkeypatch.setenv("SS_NARROW_CHUNK", "500")     # 6 chunks of R
    monkeypatch.setenv("SS_COL_FROM", "5")           # also the pattern-only wide cases stay on this kernel
    for cg in ("1", "3"):
        monkeypatch.setenv("SS_COL_CG", cg)
        w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
        got = w.spmm(R.astype(dtype))
        assert "spmm_colgroup" in ss.path_last()
        assert_close_signed(got, want, dtype)
        assert np.array_equal(got, w.spmm(R.astype(dtype)))
    monkeypatch.setenv("SS_COL", "0")
    monkeypatch.delenv("SS_COL_FROM")
    assert_close_signed(ss.DeviceSpMat(W.astype(dtype), dtype=dtype).spmm(R.astype(dtype)), want, dtype)


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("B,dtype", [(3, np.float32), (4, np.float32), (5, np.float32), (8, np.float32), (12, np.float32), (16, np.float32), (17, np.float32), (28, np.float32),
                                     (32, np.float32), (33, np.float32), (52, np.float32), (64, np.float32),
                                     (2, np.float64), (3, np.float64), (4, np.float64), (5, np.float64), (8, np.float64), (9, np.float64), (16, np.float64), (23, np.float64),
                                     (32, np.float64)])
def test_spmm_csell_lane_per_row(B, dtype, binary, monkeypatch):
    """Mid width since round 3 (fp32 B <= 64, fp64 B <= 32): the lane-per-row kernel on the compact sliced-ELL operand (spmm_csell.hip,
    csell_build in assemble.hip).  Row count no multiple of 64, empty rows, a FULL row (a 500-entry sub-row in every
    chunk: 250 steps in a block whose other lanes have ~3), an odd-length tail in most sub-rows (pad entry -> zero tile
    row), 6 chunks of R, widths that need the padded copy of R and scalar stores (5, 17, 33), several cuts (slices per
    workgroup, chunk groups: partial sums in fixed order, row sets of a wave beyond the workgroup's slices, blocks past the
    end of a wave's list).  Against scipy, against the 2-D kernel to rounding, bitwise repeatable, and every cut must
    give the same bits when it has one chunk group (the entry order is the operand's, not the cut's)."""
    rng = np.random.default_rng(300 + B)
    M, K = 9011, 2900
    W = sp.random(M, K, density=0.012, format="lil", random_state=rng, dtype=np.float64)
    W[7, :] = 1.0
    W[100:130, :] = 0.0
    W[M - 1, :] = 0.0
    W[M - 1, K - 1] = 1.0
    W = W.tocsr()
    W.data = np.ones(W.nnz) if binary else rng.random(W.nnz) + 0.5
    W.eliminate_zeros()
    R = rng.standard_normal((K, B))
    want = W @ R
    monkeypatch.setenv("SS_NARROW_CHUNK", "500")
    monkeypatch.setenv("SS_COL_FROM", "5")           # pattern-only wide cases too
    one_group = []
    for cut in (None, "3,1", "7,2", "1,6", "16,1"):
        if cut:
            monkeypatch.setenv("SS_CSELL_CUT", cut)
        w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
        got = w.spmm(R.astype(dtype))
        assert "spmm_csell" in ss.path_last(), ss.path_last()
        assert_close_signed(got, want, dtype)
        assert np.array_equal(got, w.spmm(R.astype(dtype)))
        if cut and cut.endswith(",1"):
            one_group.append(got)
    assert np.array_equal(one_group[0], one_group[1])
    monkeypatch.delenv("SS_CSELL_CUT")
    monkeypatch.setenv("SS_CSELL", "0")
    w0 = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
    other = w0.spmm(R.astype(dtype))
    assert ("spmm_colgroup" if B >= 5 else "spmm_chunked_narrow") in ss.path_last()
    assert_close_signed(other, want, dtype)


def test_spmm_csell_falls_back_to_the_2d_kernel_when_the_operand_does_not_fit(monkeypatch):
    """More than 2^24 (chunk, slice) blocks: csell_build declines (no error), the 2-D kernel serves the call."""
    rng = np.random.default_rng(9)
    M, K, nnz = 520_000, 40_000, 60_000
    W = sp.csr_matrix((rng.random(nnz) + 0.5, (rng.integers(0, M, nnz), rng.integers(0, K, nnz))), shape=(M, K))
    W.sum_duplicates()
    R = rng.standard_normal((K, 8)).astype(np.float32)
    monkeypatch.setenv("SS_NARROW_CHUNK", "16")          # 2500 chunks x 8125 slices
    w = ss.DeviceSpMat(W.astype(np.float32), dtype=np.float32)
    got = w.spmm(R)
    assert "spmm_colgroup" in ss.path_last() and "spmm_csell" not in ss.path_last(), ss.path_last()
    want = W @ R.astype(np.float64)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmm_narrow_kernel_still_serves_b3_b4_when_asked(dtype, monkeypatch):
    """SS_CSELL_ROW16=0: the narrow kernel at B = 3, 4 (by default they run on the lane-per-row kernel since round 3)."""
    monkeypatch.setenv("SS_CSELL_ROW16", "0")
    monkeypatch.setenv("SS_CSELL_B12", "0")
    rng = np.random.default_rng(4)
    W = sp.random(2111, 5000, density=0.01, format="csr", random_state=rng, dtype=np.float64)
    W.data = rng.random(W.nnz) + 0.5
    w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
    for B in (2, 3, 4):
        R = rng.standard_normal((5000, B))
        assert_close_signed(w.spmm(R.astype(dtype)), W @ R, dtype)
        assert "spmm_chunked_narrow" in ss.path_last()


def test_spmm_csell_tiny_and_degenerate_shapes():
    """One row, one column, an all-zero matrix, fewer rows than a slice, K smaller than a chunk."""
    rng = np.random.default_rng(5)
    for M, K, dens in ((1, 1, 1.0), (1, 300, 0.5), (63, 7, 0.3), (65, 2561, 0.01), (200, 40, 0.0)):
        W = sp.random(M, K, density=dens, format="csr", random_state=rng, dtype=np.float64)
        W.data = rng.random(W.nnz) + 0.5
        for B in (8, 16, 40, 64):
            R = rng.standard_normal((K, B))
            w = ss.DeviceSpMat(W.astype(np.float32), dtype=np.float32)
            got = w.spmm(R.astype(np.float32))
            if W.nnz:                                  # (an empty matrix counts as pattern-only: wide B goes to the SELL kernel)
                assert "spmm_csell" in ss.path_last()
            want = W @ R
            assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0)


@pytest.mark.parametrize("dtype,B", [(np.float32, 5), (np.float32, 7), (np.float32, 9), (np.float32, 16), (np.float32, 24),
                                     (np.float32, 50), (np.float64, 5), (np.float64, 12), (np.float64, 31)])
def test_spmm_width_routing(dtype, B, monkeypatch):
    """Row-major operands: B <= 4 narrow kernel, 5 <= B (B * sizeof <= 256 bytes) the lane-per-row kernel (fp32) or the 2-D
    kernel (under SS_CSELL=0), wider the SELL kernel.
    Widths that cannot be staged in 16-byte pieces (B = 5, 7, 9, 31; a leading dimension that is no multiple of 16
    bytes) go through the padded copy of R; SS_COL=0 (SELL kernel) must agree to rounding; results are bitwise
    repeatable (fixed summation order)."""
    rng = np.random.default_rng(B)
    M, K = 30011, 2900
    W = sp.random(M, K, density=0.012, format="csr", random_state=rng, dtype=np.float64)
    W.data = rng.random(W.nnz) + 0.5
    R = rng.standard_normal((K, B))
    want = W @ R
    monkeypatch.setenv("SS_NARROW_CHUNK", "600")     # 5 chunks of R
    w = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
    got = w.spmm(R.astype(dtype))
    assert "spmm_csell" in ss.path_last()
    assert_close_signed(got, want, dtype)
    assert np.array_equal(got, w.spmm(R.astype(dtype)))
    for b, name in ((4, "spmm_csell"), (1, "spmm_chunked_narrow"), (80, "spmm_sell")):
        Rb = rng.standard_normal((K, b))
        assert_close_signed(w.spmm(Rb.astype(dtype)), W @ Rb, dtype)
        assert any(name in k for k in ss.path_last())
    monkeypatch.setenv("SS_COL", "0")
    w0 = ss.DeviceSpMat(W.astype(dtype), dtype=dtype)
    assert_close_signed(w0.spmm(R.astype(dtype)), want, dtype)
    assert any("spmm_sell" in k for k in ss.path_last())


@pytest.mark.parametrize("weighted_y", [False, True])
def test_stage2_two_piece_tile_rows_variant(weighted_y, monkeypatch):
    """SS_SELL_QT=8 (opt-in): 32-byte tile rows = two 16-byte pieces per tile row, eight slot classes, the
    lanes of an LDS cycle split by the piece they read first, accumulators un-rotated at the store.  Several chunks
    (F is added to chunk by chunk), a query count that is no multiple of 8, an empty target for clean!."""
    Xq, Xs, Ys = O.synth_bipartite(203, 1500, 1500, 700, 0.05, 0.03, seed=77, dtype=np.float32)
    Ys = Ys.tolil(); Ys[:, 9] = 0; Ys = Ys.tocsr()
    if weighted_y:
        Ys.data = (0.5 + np.random.default_rng(3).random(Ys.nnz)).astype(np.float32)
    f64 = [m.astype(np.float64) for m in (Xq, Xs, Ys)]
    want = O.predict_factored(*f64)
    want[:, 9] = -99.0
    monkeypatch.setenv("SS_SELL_QT", "8")
    for chunk in (None, "400"):
        if chunk:
            monkeypatch.setenv("SS_SELL_CHUNK", chunk)
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
        got = g.predict("query", clean=True)
        assert_close(np.where(want == -99, 0, got), np.where(want == -99, 0, want), np.float32)
        assert ((want == -99) == (got == -99)).all()
        g.close()


def test_power_law_graph_predict_with_sorted_split_operand(monkeypatch):
    rng = np.random.default_rng(12)
    ns, nt, nq = 600, 500, 77
    deg = np.minimum((np.arange(1, ns + 1) ** -1.2) / np.mean(np.arange(1, ns + 1) ** -1.2) * 30, nt).astype(int) + 1
    rng.shuffle(deg)
    pt = np.arange(1, nt + 1) ** -1.2; pt /= pt.sum(); pt = pt[rng.permutation(nt)]
    rows = np.repeat(np.arange(ns), deg)
    cols = rng.choice(nt, size=rows.size, p=pt)
    Y = sp.csr_matrix((np.ones(rows.size), (rows, cols)), shape=(ns, nt)); Y.data[:] = 1.0
    Y[:, 3] = 0; Y.eliminate_zeros()   # one empty target for clean!
    Xq, Xs, _ = O.synth_bipartite(nq, ns, ns, nt, 0.06, 0.01, seed=4, dtype=np.float64)
    want = O.predict_factored(Xq, Xs, Y)
    kt = np.asarray((Y != 0).sum(0)).ravel()
    assert kt[3] == 0 and kt.max() > 300          # an empty target and a hot one
    want[:, kt == 0] = -99.0
    for force in ("1", "0"):
        monkeypatch.setenv("SS_SELL_SORT", force)
        monkeypatch.setenv("SS_SELL_LMAX", "256")
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Y, dtype=np.float64)
        assert_close(g.predict("query", clean=True), want, np.float64)


# ----------------------------------------------------------------------------- BASELINE config 2 at full size
def test_config2_full_size_every_row_and_linearity():
    nq = ns = nf = nt = 10_000
    Xq, Xs, Ys = O.synth_bipartite(nq, ns, nf, nt, 0.05, 0.01, seed=20250222 + 2, weighted=True, dtype=np.float32)
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    got = g.predict("query")
    assert got.shape == (nq, nt) and np.isfinite(got).all()
    f64 = [m.astype(np.float64) for m in (Xq, Xs, Ys)]
    # EVERY one of the 10^8 scores against the C restatement (fp64, OpenMP; ~1 s on the box's host cores), block by block;
    # tolerance: 1e-5 of the block's largest score (north_star), structural zeros exact
    prep = c_oracle.Prepared(*f64)
    worst = 0.0
    for r0 in range(0, nq, 2000):
        want = prep.predict(r0, r0 + 2000)
        blk = got[r0:r0 + 2000]
        assert ((want == 0) <= (blk == 0)).all()
        worst = max(worst, float(np.abs(blk - want).max() / np.abs(want).max()))
    prep.close()
    assert worst < 1e-5, worst
    # size-independent properties: scores are linear in the query's feature weights ...
    Xq2 = Xq.copy(); Xq2.data *= np.float32(0.5)
    g2 = ss.DeviceGraph.from_sparse(Xq2[:512], Xs, Ys, dtype=np.float32)
    np.testing.assert_allclose(g2.predict("query"), 0.5 * got[:512], rtol=2e-6, atol=0)
    # ... and resource is conserved: every query's scores sum to sum_s T[q,s] * (#targets of s)
    T = O.transfer_factored(f64[0][:64], f64[1], f64[2])
    np.testing.assert_allclose(got[:64].astype(np.float64).sum(1), T @ np.asarray((f64[2] != 0).sum(1)).ravel(), rtol=1e-5)


def _literal_kfold(S, Yarr, fold, k, alpha, weighted):
    """The reference's k-fold loop, literally (src/core.jl:148-201 construct(y, X, queries) -> :402-423 predict ->
    :478-484 clean!) on the CPU oracle: featurize S, then per fold the dense block graph and A*(W*W)."""
    n, nt = Yarr.shape
    names = [f"d{i:03d}" for i in range(n)]; tn = [f"t{i}" for i in range(nt)]
    Xn = O.featurize(O.Named(np.asarray(S, dtype=np.float64), names, names), float(alpha), weighted)
    Yn = O.Named(np.asarray(Yarr, dtype=np.float64), names, tn)
    want = np.zeros((n, nt))
    for phi in range(k):
        idx = [i for i in range(n) if fold[i] == phi]
        if not idx:
            continue
        members = [names[i] for i in idx]
        A, B = O.construct_queries(Yn, Xn, members)
        yq = Yn.sub(members, tn)
        yh = O.predict(A, B, yq); O.clean(yh, A, yq)
        want[idx] = yh.array
    return want



# ----------------------------------------------------------------------------- dense-similarity regime (MFMA stage 1)
@pytest.mark.parametrize("engine", ["bf16-planes", "bf16-planes-ring", "fp32-mfma"])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("shape", [(70, 333, 41), (1, 64, 5), (130, 129, 300)])
def test_dense_similarity_path_query_and_loo(shape, weighted, engine, monkeypatch):
    """Both stage-1 engines of the dense regime: the default bf16 MFMA on exact bf16 planes of the fp32 operands
    (dense_bf16.hip: 3 plane products unweighted, 6 weighted; 128 x 128 kernel and the 256 x 256 ring kernel) and the
    fp32-input MFMA kernel (SS_DENSE_BF16=0)."""
    if engine == "fp32-mfma":
        monkeypatch.setenv("SS_DENSE_BF16", "0")
    if engine == "bf16-planes-ring":     # the 256 x 256 ring kernel that large shapes pick by themselves
        monkeypatch.setenv("SS_DENSE_RING", "1")
    nq, ns, nt = shape
    rng = np.random.default_rng(ns)
    Ss = rng.random((ns, ns)).astype(np.float32); Ss = ((Ss + Ss.T) / 2).astype(np.float32); np.fill_diagonal(Ss, 1.0)
    Sq = rng.random((nq, ns)).astype(np.float32)
    Y = sp.random(ns, nt, density=0.08, format="csr", random_state=rng, dtype=np.float32); Y.data[:] = 1.0
    alpha = np.float32(0.35)   # ~65 % fill: far denser than the CSR path is meant for
    Ss[3, 7] = Ss[7, 3] = alpha  # exactly alpha: kept
    g = ss.DeviceGraph.from_similarity(Sq, Ss, Y, alpha=float(alpha), weighted=weighted)
    Xq = O.cutoff(Sq.astype(np.float64), float(alpha), weighted)
    Xs = O.cutoff(Ss.astype(np.float64), float(alpha), weighted)
    Y64 = Y.astype(np.float64)
    kf, ks, kt = g.degrees()
    okf, oks, okt = O.degrees(Xs, Y64)
    np.testing.assert_array_equal(kf, okf); np.testing.assert_array_equal(ks, oks); np.testing.assert_array_equal(kt, okt)
    assert_close(g.predict("query"), O.predict_factored(Xq, Xs, Y64), np.float32)
    want = O.predict_loo_factored(Xs, Y64, clean_flag=True)
    assert_close(g.predict_loo(clean=True), want, np.float32)
    lo, hi = ns // 3, ns - 1
    assert_close(g.predict_loo(lo, hi, clean=True, layout="col"), want[lo:hi], np.float32)
    # source rows (= predict(A, ytrain), src/core.jl:446-466): feature path on the matrix cores + sparse target path,
    # against the CPU oracle's source-row form (not against another device path)
    want_src = O.predict_factored(None, Xs, Y64, rows="source")
    assert_close(g.predict("source"), want_src, np.float32)
    want_clean = want_src.copy(); want_clean[:, okt == 0] = -99.0        # clean!: targets without any edge in A (src/core.jl:479)
    assert_close(g.predict("source", lo, hi, clean=True), want_clean[lo:hi], np.float32)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("shape", [(70, 333, 41), (1, 64, 5), (130, 257, 300)])
def test_dense_similarity_path_fp64(shape, weighted):
    """The dense-similarity regime in the reference's default precision (Float64, src/core.jl:402 GPU=false): fp64 matrix
    instruction (dense_f64.hip), cutoff fused into the operand staging -- query rows, source rows, leave-one-out (rank-1
    degree corrections) and k-fold (member rows gathered, per-fold degrees), all to 1e-12."""
    nq, ns, nt = shape
    rng = np.random.default_rng(ns + 1)
    Ss = rng.random((ns, ns)); Ss = (Ss + Ss.T) / 2; np.fill_diagonal(Ss, 1.0)
    Sq = rng.random((nq, ns))
    Y = sp.random(ns, nt, density=0.08, format="csr", random_state=rng, dtype=np.float64); Y.data[:] = 1.0
    alpha = 0.35
    Ss[3, 7] = Ss[7, 3] = alpha  # exactly alpha: kept
    g = ss.DeviceGraph.from_similarity(Sq, Ss, Y, alpha=alpha, weighted=weighted, dtype=np.float64)
    Xq, Xs = O.cutoff(Sq, alpha, weighted), O.cutoff(Ss, alpha, weighted)
    kf, ks, kt = g.degrees()
    okf, oks, okt = O.degrees(Xs, Y)
    np.testing.assert_array_equal(kf, okf); np.testing.assert_array_equal(ks, oks); np.testing.assert_array_equal(kt, okt)
    assert_close(g.predict("query"), O.predict_factored(Xq, Xs, Y), np.float64)
    assert "transfer_dense_f64_mfma" in ss.path_last()
    want = O.predict_loo_factored(Xs, Y, clean_flag=True)
    assert_close(g.predict_loo(clean=True), want, np.float64)
    lo, hi = ns // 3, ns - 1
    assert_close(g.predict_loo(lo, hi, clean=True, layout="col"), want[lo:hi], np.float64)
    assert_close(g.predict("source"), O.predict_factored(None, Xs, Y, rows="source"), np.float64)
    folds = rng.integers(0, 4, ns).astype(np.int32)
    # k-fold against the reference's literal fold loop on the CPU oracle (construct -> predict -> clean! per fold)
    assert_close(g.predict_kfold(folds, 4, clean=True), _literal_kfold(Ss, Y.toarray(), folds, 4, alpha, weighted), np.float64)


def test_dense_similarity_equals_sparse_path_on_the_same_input():
    rng = np.random.default_rng(77)
    ns, nq, nt = 500, 100, 64
    Ss = rng.random((ns, ns)).astype(np.float32); Sq = rng.random((nq, ns)).astype(np.float32)
    Y = (rng.random((ns, nt)) < 0.05).astype(np.float32)
    a = ss.DeviceGraph.from_similarity(Sq, Ss, sp.csr_matrix(Y), alpha=0.8, weighted=True).predict("query")
    b = ss.DeviceGraph.from_dense(Sq, Ss, Y, alpha=np.float32(0.8), weighted=True, dtype=np.float32).predict("query")
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-9)


# ----------------------------------------------------------------------------- k-fold in one call
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_kfold_equals_the_reference_fold_loop(dtype, monkeypatch):
    rng = np.random.default_rng(21)
    n, nt, k = 60, 17, 5
    S = rng.random((n, n)); S = (S + S.T) / 2; np.fill_diagonal(S, 1.0)
    names = [f"d{i:02d}" for i in range(n)]; tn = [f"t{i}" for i in range(nt)]
    Xn = O.featurize(O.Named(S, names, names), 0.6, True)
    Yarr = (rng.random((n, nt)) < 0.2).astype(float)
    Yarr[:, 3] = 0; Yarr[7, 3] = 1; Yarr[9, 3] = 1   # a target whose only edges are in one fold -> clean!
    Yn = O.Named(Yarr, names, tn)
    fold = rng.integers(0, k, size=n).astype(np.int32); fold[7] = fold[9] = 2
    want = np.zeros((n, nt))
    for phi in range(k):
        members = [names[i] for i in range(n) if fold[i] == phi]
        A, B = O.construct_queries(Yn, Xn, members)
        yq = Yn.sub(members, tn)
        yh = O.predict(A, B, yq); O.clean(yh, A, yq)
        want[[i for i in range(n) if fold[i] == phi]] = yh.array
    assert (want[7] == -99).any()
    for force_sorted in ("0", "1"):
        monkeypatch.setenv("SS_SELL_SORT", force_sorted)
        g = ss.DeviceGraph.from_dense(None, Xn.array.astype(dtype), Yarr.astype(dtype), dtype=dtype)
        assert_close(g.predict_kfold(fold, k, clean=True), want, dtype)
    # one source per fold is leave-one-out
    monkeypatch.delenv("SS_SELL_SORT")
    g = ss.DeviceGraph.from_dense(None, Xn.array.astype(dtype), Yarr.astype(dtype), dtype=dtype)
    assert_close(g.predict_kfold(np.arange(n, dtype=np.int32), n, clean=True), g.predict_loo(clean=True), dtype)


# ----------------------------------------------------------------------------- ranked evaluation on the device
def test_topl_matches_stable_descending_sort_with_ties():
    rng = np.random.default_rng(17)
    x = np.round(rng.random((37, 5000)).astype(np.float32) * 50) / 50      # heavy ties
    x[3, :] = 0.25                                                         # a whole row of ties
    x[5, 100:140] = -99.0
    for L in (1, 20, 64, 333, 1024):
        idx, val = ss.topl(x, L)
        want = np.argsort(-x, axis=1, kind="stable")[:, :L]
        np.testing.assert_array_equal(idx, want)
        np.testing.assert_array_equal(val, np.take_along_axis(x, want, 1))


def test_topl_signed_zeros_rank_as_julia_isless_orders_them(monkeypatch):
    """sortperm(yhat, rev=true) compares with isless, for which -0.0 < +0.0: +0.0 columns come first (ascending), then the
    -0.0 columns (include/simspread_hip.h, ss_topl_f32).  numpy's argsort(-x) calls the two zeros equal -- the difference
    tools/fuzz_spmm_topl.py stumbled over; SimSpread scores hold no -0.0."""
    x = np.zeros((2, 3000), np.float32)
    x[0, ::2] = -0.0
    x[0, 5] = 1.0
    x[1, 1000:] = -0.0
    x[1, 7] = -1.0
    odd = [c for c in range(1, 3000, 2) if c != 5]
    for bound in ("1", "0"):                            # fast path and radix select alone
        monkeypatch.setenv("SS_TOPL_BOUND", bound)
        idx, val = ss.topl(x, 1024)
        assert idx[0].tolist() == [5] + odd[:1023]
        assert idx[1].tolist() == [c for c in range(1000) if c != 7] + list(range(1000, 1025))
        assert not np.signbit(val[1, :999]).any() and np.signbit(val[1, 999:]).all()


def test_topl_bound_path_and_radix_fallback_agree(monkeypatch):
    """ss_topl_f32 has a fast path (lower bound from per-thread maxima, candidates sorted) and a radix-select fallback for
    rows with more than 4096 candidates: distinct scores stay on the fast path, long runs of equal scores at the bound
    overflow it; both must give the order of a stable descending sort, on rows wider than one pass of the workgroup."""
    rng = np.random.default_rng(23)
    x = rng.standard_normal((24, 70001)).astype(np.float32)
    x[1, :] = np.float32(0.0)                          # every element ties: fallback
    x[2, 5000:] = np.float32(-99.0)                    # cleaned columns below the bound
    x[3, ::2] = np.float32(3.0)                        # 35001 ties above everything else: fallback (L <= ties)
    x[4, :90] = np.float32(7.0)                        # ties inside the top L, distinct scores at the bound
    x[5] = np.round(x[5] * 4) / 4                      # a few dozen distinct values: ties at the bound
    for L in (1, 100, 1024):
        idx, val = ss.topl(x, L)
        want = np.argsort(-x, axis=1, kind="stable")[:, :L]
        np.testing.assert_array_equal(idx, want)
        np.testing.assert_array_equal(val, np.take_along_axis(x, want, 1))
    monkeypatch.setenv("SS_TOPL_BOUND", "0")           # radix select alone
    idx0, val0 = ss.topl(x, 100)
    np.testing.assert_array_equal(idx0, np.argsort(-x, axis=1, kind="stable")[:, :100])


def test_recall_precision_at_L_reference_kat_and_device_scores(kats):
    k = kats["at_L"]
    for case in k["cases"]:
        assert ss.recallatL(k["y"], k["yhat"], k["grouping"], case["L"]) == pytest.approx(case["recall"])
        assert ss.precisionatL(k["y"], k["yhat"], k["grouping"], case["L"]) == pytest.approx(case["precision"])
    # scores stay on the device: top-20 of each leave-one-out row, hits from the labels
    import torch
    ss.use_torch_stream()
    rng = np.random.default_rng(2)
    n, nt = 400, 300
    S = rng.random((n, n)).astype(np.float32); S = (S + S.T) / 2; np.fill_diagonal(S, 1.0)
    Y = (rng.random((n, nt)) < 0.05).astype(np.float32)
    g = ss.DeviceGraph.from_dense(None, S, Y, alpha=np.float32(0.7), weighted=True, dtype=np.float32)
    out = torch.empty((n, nt), dtype=torch.float32, device="cuda")
    g.predict_loo(clean=True, out=out)
    idx, _ = ss.topl(out, 20)
    host = out.cpu().numpy()
    np.testing.assert_array_equal(idx.cpu().numpy(), np.argsort(-host, axis=1, kind="stable")[:, :20])
    hits = np.take_along_axis(Y, idx.cpu().numpy().astype(np.int64), 1).sum(1)
    grouping = np.repeat(np.arange(n), nt)
    assert ss.precisionatL(Y.ravel(), host.ravel(), grouping, 20) == pytest.approx(hits.mean() / 20)


def test_timing_hold_accumulates_and_stream_ordered_predictions():
    """ss_timing_hold: stage timings of successive calls add up; device row-major predictions are enqueued in
    stream order (the second call may start before the host has seen the first finish) and still agree."""
    import torch
    rng = np.random.default_rng(9)
    nq, ns, nt = 300, 400, 90
    Xq = sp.random(nq, ns, density=0.1, format="csr", random_state=rng, dtype=np.float32)
    Xs = sp.random(ns, ns, density=0.1, format="csr", random_state=rng, dtype=np.float32)
    Ys = sp.random(ns, nt, density=0.05, format="csr", random_state=rng, dtype=np.float32); Ys.data[:] = 1.0
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    want = g.predict("query")                       # host result: complete on return
    a = torch.empty((nq, nt), dtype=torch.float32, device="cuda")
    b = torch.empty((nq, nt), dtype=torch.float32, device="cuda")
    ss.timing_hold(True)
    for _ in range(3):
        g.predict("query", out=a)
        g.predict("query", out=b)
    t = ss.timing_last()
    ss.timing_hold(False)
    assert t["spmm_launches"] == 6 and t["transfer_launches"] == 6
    assert t["spmm_ms"] > 0 and t["total_ms"] >= t["spmm_ms"] + t["transfer_ms"] - 1e-6
    np.testing.assert_array_equal(a.cpu().numpy(), want)
    np.testing.assert_array_equal(b.cpu().numpy(), want)
    g.predict("query", out=a)
    assert ss.timing_last()["spmm_launches"] == 1   # back to per-call timings


# ----------------------------------------------------------------------------- similarity producer (tutorial step)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_jaccard_similarity_reproduces_the_reference_iris_simmat(dtype):
    """docs/src/tutorial/fishers-flowers.jl:66 on docs/src/tutorial/data/iris.features must give iris.simmat
    (both files are committed fixtures of the reference: tests/golden/iris)."""
    here = os.path.join(os.path.dirname(__file__), "golden", "iris")
    def read(p):
        with open(os.path.join(here, p)) as f:
            lines = f.read().splitlines()
        return np.array([[float(v) for v in l.split()[1:]] for l in lines[1:]])
    F, want = read("iris.features"), read("iris.simmat")
    got = ss.jaccard_similarity(F, dtype=dtype)
    assert got.shape == (150, 150) and got.dtype == dtype
    np.testing.assert_allclose(got, want, rtol=1e-12 if dtype == np.float64 else 2e-6, atol=0)
    np.testing.assert_array_equal(got, got.T)
    np.testing.assert_array_equal(np.diag(got), np.ones(150, dtype))


def test_jaccard_similarity_shapes_and_zero_rows():
    import torch
    rng = np.random.default_rng(4)
    X = rng.random((203, 37))
    X[5] = 0; X[77] = 0                      # all-zero rows: similarity 1 to each other, 0 to the rest
    mn = np.minimum(X[:, None, :], X[None, :, :]).sum(-1); mx = np.maximum(X[:, None, :], X[None, :, :]).sum(-1)
    want = np.where(mx == 0, 1.0, mn / np.where(mx == 0, 1, mx))
    np.testing.assert_allclose(ss.jaccard_similarity(X), want, rtol=1e-12)
    got = ss.jaccard_similarity(torch.from_numpy(X.astype(np.float32)).cuda())
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=3e-6)
    assert ss.jaccard_similarity(np.zeros((0, 4))).shape == (0, 0)


@pytest.mark.parametrize("weighted", [False, True])
def test_dense_similarity_kfold_equals_the_reference_fold_loop(weighted):
    """k-fold in the dense regime (per-fold degree recount on the dense matrix, members' rows gathered into the
    query planes) against the reference's literal fold loop on the CPU oracle."""
    rng = np.random.default_rng(31)
    ns, nt, k = 333, 29, 4
    Ss = rng.random((ns, ns)).astype(np.float32); Ss = ((Ss + Ss.T) / 2).astype(np.float32); np.fill_diagonal(Ss, 1.0)
    Y = (rng.random((ns, nt)) < 0.08).astype(np.float32)
    Y[:, 5] = 0; Y[10, 5] = 1; Y[20, 5] = 1            # a target whose edges all sit in one fold -> clean!
    fold = rng.integers(0, k, size=ns).astype(np.int32); fold[10] = fold[20] = 1
    alpha = 0.6
    dense = ss.DeviceGraph.from_similarity(None, Ss, sp.csr_matrix(Y), alpha=alpha, weighted=weighted)
    # the reference's literal fold loop on the CPU oracle (thresholded at the fp32 alpha the device compares with)
    want = _literal_kfold(Ss, Y, fold, k, np.float32(alpha), weighted)
    got = dense.predict_kfold(fold, k, clean=True)
    # ... and the sparse device path on the same thresholded input agrees too (two device paths, one oracle)
    sparse = ss.DeviceGraph.from_dense(None, Ss, Y, alpha=np.float32(alpha), weighted=weighted, dtype=np.float64)
    assert_close(sparse.predict_kfold(fold, k, clean=True), want, np.float64)
    assert ((want == -99) == (got == -99)).all() and (want[10] == -99).any()
    assert_close(got, want, np.float32)


def test_tutorial_example_runs_end_to_end_on_the_device():
    import importlib.util
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "iris_tutorial.py")
    spec = importlib.util.spec_from_file_location("iris_tutorial", p)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    out = mod.main(0.9)
    for weighted in (True, False):
        assert out[weighted]["AuROC"] > 0.9 and out[weighted]["accuracy_of_predicted"] > 0.9   # iris is easy


def test_stage1_chunk_group_order_is_bitwise_the_old_order(monkeypatch):
    """Round 3: with more than 8 column chunks stage 1 walks the groups of 8 chunks one after the other over all rows
    (working set of the XCDs' L2 / Infinity Cache) instead of giving every row all its chunks in turn.  Only the launch
    order changes: query rows, source rows and leave-one-out rows must be bit-identical to SS_TRANSFER_ORDER=0, and right."""
    Xq, Xs, Ys = O.synth_bipartite(131, 3000, 3000, 400, 0.04, 0.02, seed=8, dtype=np.float32)
    monkeypatch.setenv("SS_TRANSFER_CHUNK", "100")          # 30 chunks -> 32: four groups of 8
    res = {}
    for order in ("1", "0"):
        monkeypatch.setenv("SS_TRANSFER_ORDER", order)
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
        g3 = ss.DeviceGraph.from_sparse(None, Xs, Ys, dtype=np.float32)
        res[order] = (g.predict("query").copy(), g.predict("source", 10, 300).copy(), g3.predict_loo(500, 900, clean=True).copy())
        g.close(); g3.close()
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)
    assert_close(res["1"][0], O.predict_factored(*(m.astype(np.float64) for m in (Xq, Xs, Ys))), np.float32)


# ----------------------------------------------------------------------------- stage-1 variants of round 3 (opt-in kernels)
@pytest.mark.parametrize("variant", [
    {"SS_TRANSFER_V": "2", "SS_TRANSFER_FIX": "0"},                    # query-block workgroups, plain read-add-write
    {"SS_TRANSFER_V": "2"},                                            # ... with fixed-point ds_add_u32 sums
    {"SS_TRANSFER_V": "2", "SS_TRANSFER_QFLAT": "4"},                  # flat stream of quads
    {"SS_TRANSFER_V": "2", "SS_TRANSFER_WIDE": "4"},                   # 16 bytes per lane, step generator
    {"SS_TRANSFER_V": "2", "SS_TRANSFER_WIDE2": "1"},                  # 16 bytes per lane, static 20-step schedule
    {"SS_TRANSFER_V": "1", "SS_TRANSFER_FIX1": "1"},                   # single-wave kernel with fixed-point sums
    {"SS_TRANSFER_V": "1", "SS_TRANSFER_LD": "1"},                     # single-wave kernel with buffer loads
])
@pytest.mark.parametrize("chunk,dx", [(None, 0.05), ("3000", 0.05), ("500", 0.2)])
def test_stage1_variants_match_the_oracle(variant, chunk, dx, monkeypatch):
    """The measured-but-not-default stage-1 kernels against the CPU oracle: default chunking
    (short sub-rows), one 3000-column chunk (150-entry sub-rows: second and third levels of the wide kernels, their
    per-lane rest path past 96 entries) and 20 % fill in 500-column chunks (EVERY sub-row ~100 entries: the lists of the
    second and third levels overflow in every group of 64 features -- the case tools/fuzz_predict.py found a double count
    in).  Zero-degree features, an empty query row, weighted and unweighted features.
    Fixed-point sums: error bound n_terms * 2^-30 * (sum |coefficients|) * max |value| per score."""
    for k, v in variant.items():
        monkeypatch.setenv(k, v)
    if chunk:
        monkeypatch.setenv("SS_TRANSFER_CHUNK", chunk)
    for weighted in (True, False):
        Xq, Xs, Ys = O.synth_bipartite(257, 3000, 3000, 200, dx, 0.03, seed=11, weighted=weighted, dtype=np.float32)
        Xq = Xq.tolil(); Xq[5, :] = 0; Xq = Xq.tocsr()                     # a query without features
        Xs = Xs.tolil(); Xs[:, 17] = 0; Xs[:, 18] = 0; Xs = Xs.tocsr()     # features nobody has (degree 0)
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
        got = g.predict("query")
        path = ss.path_last()
        if variant.get("SS_TRANSFER_V") == "2":
            assert "transfer_block" in path, path
            for tag, key in (("qflat", "SS_TRANSFER_QFLAT"), ("wide", "SS_TRANSFER_WIDE"), ("wide2", "SS_TRANSFER_WIDE2")):
                if key in variant:
                    assert tag in path, path
        want = O.predict_factored(Xq.astype(np.float64), Xs.astype(np.float64), Ys.astype(np.float64))
        assert_close(got, want, np.float32)
        assert (got[5] == 0).all()
        g.close()
