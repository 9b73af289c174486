"""The oracle against the reference's own known-answer tests (test/runtests.jl) and
against itself (literal dense form == factored sparse form == LOO identity)."""
import os

import numpy as np
import pytest

from oracle import simspread_oracle as O
from oracle.simspread_oracle import Named


def test_k_kat(kats):
    M = np.array(kats["k"]["M"], dtype=float)
    assert int(O.k(M[0])) == 0
    assert O.k(M).ravel().tolist() == kats["k"]["row_degrees"]
    assert O.k(M).shape == (4, 1)


def test_cutoff_kats(kats):
    c = kats["cutoff"]
    x, y, z = c["x"], np.array(c["y"]).reshape(-1, 1), np.array(c["z"])
    for case in c["cases"]:
        a = case["alpha"]
        assert O.cutoff(x, a, False) == pytest.approx(case["x_bin"])
        assert O.cutoff(x, a, True) == pytest.approx(case["x_w"])
        np.testing.assert_allclose(O.cutoff(y, a, False).ravel(), case["y_bin"])
        np.testing.assert_allclose(O.cutoff(y, a, True).ravel(), case["y_w"])
        np.testing.assert_allclose(O.cutoff(z, a, False), case["z_bin"])
        np.testing.assert_allclose(O.cutoff(z, a, True), case["z_w"])


def test_cutoff_inclusive():
    assert O.cutoff(0.5, 0.5, False) == 1.0
    assert O.cutoff(np.nextafter(0.5, 0), 0.5, False) == 0.0


def test_featurize_kat(kats):
    f = kats["featurize"]
    M0 = Named(f["M0"], f["rows"], f["cols"])
    u = O.featurize(M0, f["alpha"], False)
    w = O.featurize(M0, f["alpha"], True)
    assert u.cols == f["out_cols"] and w.cols == f["out_cols"]
    np.testing.assert_array_equal(u.array, f["unweighted"])
    np.testing.assert_array_equal(w.array, f["weighted"])


def test_construct_kat(kats):
    c = kats["construct"]
    X = Named(c["X"], c["X_rows"], c["X_cols"])
    y = Named(c["y"], c["y_rows"], c["y_cols"])
    A, B = O.construct_queries(y, X, c["queries"])
    assert A.rows == c["node_order"] and A.cols == c["node_order"]
    assert B.rows == c["node_order"] and B.cols == c["node_order"]
    np.testing.assert_array_equal(A.array, A.array.T)
    assert not B.array[0].any() and not B.array[:, 0].any()
    X2 = Named(c["X"], c["X_rows"], c["X_rows"])  # features named like sources
    with pytest.raises(AssertionError, match=c["same_names_message"]):
        O.construct_queries(y, X2, c["queries"])
    X3 = Named(c["row_mismatch_X"], c["row_mismatch_rows"], c["X_cols"])
    with pytest.raises(AssertionError, match=c["row_mismatch_message"]):
        O.construct_queries(y, X3, c["queries"])


def test_spread_kat(kats):
    s = kats["spread"]
    # Julia's `≈` on arrays is norm-based: norm(x-y) <= rtol*max(norm(x), norm(y))
    got, want = O.spread(np.array(s["M"], float)), np.array(s["W"])
    assert np.linalg.norm(got - want) <= s["rtol"] * max(np.linalg.norm(got), np.linalg.norm(want))
    np.testing.assert_allclose(got, [[1, 0, 0], [0.5, 0.5, 0], [1 / 3, 1 / 3, 1 / 3]], rtol=1e-15)


def test_spread_zero_degree_rows_give_zero():
    W = O.spread(np.array([[0.0, 0.0], [2.0, 0.5]]))
    np.testing.assert_array_equal(W, [[0, 0], [1.0, 0.25]])  # count, not weight sum


def test_predict_kat_exact(kats):
    p = kats["predict"]
    A = Named(p["A"], p["names"], p["names"])
    B = Named(p["B"], p["names"], p["names"])
    y = Named(p["y"], p["y_rows"], p["y_cols"])
    yhat = O.predict(A, B, y)
    assert yhat.rows == p["y_rows"] and yhat.cols == p["y_cols"]
    assert (yhat.array == np.array(p["yhat"])).all()  # reference uses exact ==


def test_clean_kat(kats):
    c = kats["clean"]
    A = Named(c["A"], c["names"], c["names"])
    y = Named(c["y"], c["y_rows"], c["y_cols"])
    yhat = Named(c["yhat_in"], c["y_rows"], c["y_cols"])
    O.clean(yhat, A, y)
    np.testing.assert_array_equal(yhat.array, c["yhat_out"])


def _random_named(seed, nq=5, ns=17, nf=17, nt=7, weighted=True):  # Nf == Ns: the reference name check needs it (quirk 6)
    rng = np.random.default_rng(seed)
    def m(r, c, d):
        a = (rng.random((r, c)) < d) * (rng.random((r, c)) * 0.5 + 0.5 if weighted else 1.0)
        return a
    Xq, Xs, Ys = m(nq, nf, 0.4), m(ns, nf, 0.4), (rng.random((ns, nt)) < 0.3).astype(float)
    Xs[:, 2] = 0.0     # a zero-degree feature
    Xs[3, :] = 0.0     # an isolated source
    Ys[3, :] = 0.0
    Ys[:, 1] = 0.0     # a zero-degree target
    q = [f"q{i}" for i in range(nq)]
    s = [f"s{i}" for i in range(ns)]
    f = [f"f{i}" for i in range(nf)]
    t = [f"t{i}" for i in range(nt)]
    return (Named(Xq, q, f), Named(Xs, s, f), Named(Ys, s, t),
            Named(np.zeros((nq, nt)), q, t))


@pytest.mark.parametrize("seed,weighted", [(1, True), (2, False), (3, True)])
def test_literal_equals_factored(seed, weighted):
    Xq, Xs, Ys, yq = _random_named(seed, weighted=weighted)
    A, B = O.construct_split(Ys, yq, Xs, Xq)
    lit_q = O.predict(A, B, yq).array
    lit_s = O.predict(A, B, Ys).array
    np.testing.assert_allclose(O.predict_factored(Xq.array, Xs.array, Ys.array, "query"), lit_q, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(O.predict_factored(Xq.array, Xs.array, Ys.array, "source"), lit_s, rtol=1e-12, atol=1e-15)
    A3 = O.construct_single(Ys, Xs)
    np.testing.assert_allclose(O.predict_single(A3, Ys).array, lit_s, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("seed,weighted", [(4, True), (5, False)])
def test_loo_identity(seed, weighted):
    rng = np.random.default_rng(seed)
    n, nt = 24, 6
    S = rng.random((n, n)); S = (S + S.T) / 2; np.fill_diagonal(S, 1.0)
    names = [f"d{i:02d}" for i in range(n)]
    X = O.featurize(Named(S, names, names), 0.55, weighted)
    Y = Named((rng.random((n, nt)) < 0.25).astype(float), names, [f"t{i}" for i in range(nt)])
    Y.array[:, 2] = 0.0; Y.array[5, 2] = 1.0   # target whose only edge is source 5 -> clean! case
    fast = O.predict_loo_factored(X.array, Y.array, clean_flag=True)
    for i, nm in enumerate(names):
        A, B = O.construct_queries(Y, X, [nm])
        yq = Y.sub([nm], Y.cols)
        yhat = O.predict(A, B, yq)
        O.clean(yhat, A, yq)
        np.testing.assert_allclose(fast[i], yhat.array[0], rtol=1e-12, atol=1e-15)
    assert fast[5, 2] == -99.0


def test_feature_filter_strips_all_leading_f():
    # quirk: lstrip(f, 'f') removes every leading 'f' (src/core.jl:152)
    names = ["foo", "bar", "oo"]
    X = Named(np.ones((3, 3)), names, ["f" + n for n in names])
    y = Named(np.eye(3), names, ["t0", "t1", "t2"])
    A, _ = O.construct_queries(y, X, ["oo"])
    # "ffoo" -> "oo" and "foo" -> "oo": both feature columns vanish, only "fbar" stays
    assert A.rows == ["oo", "foo", "bar", "fbar", "t0", "t1", "t2"]


def _iris():
    here = os.path.join(os.path.dirname(__file__), "golden", "iris")
    def read(p):
        with open(os.path.join(here, p)) as f:
            lines = f.read().splitlines()
        cols = lines[0].split()
        rows = [l.split()[0] for l in lines[1:]]
        vals = np.array([[float(v) for v in l.split()[1:]] for l in lines[1:]])
        return rows, cols, vals
    rows, fc, F = read("iris.features")
    _, cc, C = read("iris.classes")
    mn = np.minimum(F[:, None, :], F[None, :, :]).sum(-1)
    mx = np.maximum(F[:, None, :], F[None, :, :]).sum(-1)
    return rows, cc, mn / mx, C


def test_iris_simmat_matches_reference_file():
    p = os.path.join(os.path.dirname(__file__), "golden", "iris", "iris.simmat")
    rows, _, S, _ = _iris()
    with open(p) as f:
        lines = f.read().splitlines()
    ref = np.array([[float(v) for v in l.split()[1:]] for l in lines[1:]])
    np.testing.assert_allclose(S, ref, rtol=0, atol=1e-14)


def test_iris_single_loo_fold_literal_vs_factored():
    rows, cc, S, C = _iris()
    X = O.featurize(Named(S, rows, rows), 0.9, True)
    Y = Named(C, rows, cc)
    fast = O.predict_loo_factored(X.array, Y.array, clean_flag=True, queries=[0, 77, 149])
    for o, i in enumerate([0, 77, 149]):
        A, B = O.construct_queries(Y, X, [rows[i]])
        yq = Y.sub([rows[i]], cc)
        yhat = O.predict(A, B, yq)
        O.clean(yhat, A, yq)
        np.testing.assert_allclose(fast[o], yhat.array[0], rtol=1e-12, atol=1e-15)
    assert A.array.shape == (302, 302)


def test_name_check_needs_equal_counts():
    # quirk 6: element-wise compare of the sorted name vectors -> DimensionMismatch unless Nf == Ns
    X = Named(np.ones((3, 2)), ["a", "b", "c"], ["fx", "fy"])
    y = Named(np.ones((3, 1)), ["a", "b", "c"], ["t"])
    with pytest.raises(ValueError, match="DimensionMismatch"):
        O.construct_single(y, X)


def test_loo_dense_form_equals_factored_form():
    """predict_loo_dense (used to check the dense-similarity regime at sizes where a sparse copy of a 90 % full
    matrix is impractical) against predict_loo_factored, weighted and unweighted, with an isolated target."""
    rng = np.random.default_rng(5)
    n, nt = 60, 17
    S = rng.random((n, n)); S = np.triu(S, 1); S = S + S.T; np.fill_diagonal(S, 1.0)
    Y = (rng.random((n, nt)) < 0.15).astype(np.float64)
    Y[:, 3] = 0.0
    Y[5, 4] = 1.0; Y[:5, 4] = 0.0; Y[6:, 4] = 0.0          # a target whose only edge belongs to one query
    for alpha, weighted in ((0.1, True), (0.5, False), (0.9, True)):
        X = O.cutoff(S, alpha, weighted)
        a = O.predict_loo_dense(X, Y, clean_flag=True)
        b = O.predict_loo_factored(X, Y, clean_flag=True)
        assert np.abs(a - b).max() < 1e-13
        assert ((a == -99) == (b == -99)).all()


def test_c_oracle_equals_python_oracle():
    """oracle/factored.c (the checker of the full-size GPU tests and bench.py's cpu_baseline) against the numpy
    factored form it restates, on a seeded graph with a zero-degree feature, an isolated source and weighted edges."""
    from oracle import c_oracle
    Xq, Xs, Ys = O.synth_bipartite(70, 90, 90, 41, 0.08, 0.06, seed=11, weighted=True, dtype=np.float64)
    Xs = Xs.tolil(); Xs[:, 7] = 0.0; Xs[12, :] = 0.0; Xs = Xs.tocsr(); Xs.eliminate_zeros()
    Ys = Ys.tolil(); Ys[12, :] = 0.0; Ys = Ys.tocsr(); Ys.eliminate_zeros()
    want = O.predict_factored(Xq, Xs, Ys, "query")
    got = c_oracle.predict_query(Xq, Xs, Ys)
    assert np.abs(got - want).max() <= 1e-15 * max(1.0, np.abs(want).max()) * 64
    prep = c_oracle.Prepared(Xq, Xs, Ys)
    assert np.array_equal(prep.predict(3, 40), got[3:40])       # prepared form == one-shot form, any row block
    assert np.array_equal(prep.predict(threads=1), got)
    prep.close()
