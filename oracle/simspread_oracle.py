"""CPU oracle for the SimSpread.jl hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy (fp64) restatement of what the reference computes on
the path ``featurize -> construct -> spread -> predict -> clean!``.  It exists so
the HIP kernels can be checked; it is never the thing shipped or measured.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package (``simspread.jl_amd``) must never import it.

Pinning: Julia is not installed in the build container, so the reference cannot be
run; the oracle is pinned by the reference's own known-answer tests
(``test/runtests.jl:20-26,36-81,83-118,120-183`` transcribed as data in
``tests/golden/reference_kats.json`` and asserted in ``tests/test_oracle.py``).
GPU=true results, weighted-feature prediction and ``predict(A, ytrain)`` are not
covered by any reference test: for those the literal dense restatement below *is*
the definition (parity unpinned beyond the formula), see DESIGN.md.

Two forms are provided and cross-checked against each other in the tests:

* literal  : dense block adjacency ``A``/``B``, ``W = spread(B)``, ``F = A @ W @ W``
             exactly as ``src/core.jl:148-201,217-276,308-337,365-371,402-466``.
* factored : ``Yq = (Xq D_f^-1) Xs' (D_s^-1 Ys)`` and the source-row / leave-one-out
             forms, on scipy.sparse, fp64.  This is the
             algorithmic peer of the device kernels.

All matrices here are row-major numpy arrays; names are Python lists of str.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import numpy as np

try:  # scipy is only needed by the factored forms
    import scipy.sparse as sp
except Exception:  # pragma: no cover
    sp = None


# --------------------------------------------------------------------------- names
@dataclass
class Named:
    """Minimal stand-in for NamedArrays.NamedMatrix (values + two name lists).

    The reference indexes by name everywhere (``src/core.jl:167,171-172,197-198,421``).
    Equality in the reference's tests compares values only.
    """

    array: np.ndarray
    rows: List[str]
    cols: List[str]

    def __post_init__(self):
        self.array = np.asarray(self.array, dtype=np.float64)
        if self.array.ndim != 2:
            raise ValueError("Named needs a matrix")
        self.rows = [str(r) for r in self.rows]
        self.cols = [str(c) for c in self.cols]
        if len(self.rows) != self.array.shape[0] or len(self.cols) != self.array.shape[1]:
            raise ValueError("name lists do not match matrix shape")

    def names(self, dim: int) -> List[str]:
        return list(self.rows if dim == 1 else self.cols)

    def sub(self, rows: Sequence[str], cols: Sequence[str]) -> "Named":
        ri = {n: i for i, n in enumerate(self.rows)}
        ci = {n: i for i, n in enumerate(self.cols)}
        r = [ri[str(n)] for n in rows]
        c = [ci[str(n)] for n in cols]
        return Named(self.array[np.ix_(r, c)], [str(n) for n in rows], [str(n) for n in cols])

    def copy(self) -> "Named":
        return Named(self.array.copy(), list(self.rows), list(self.cols))


# --------------------------------------------------------------------------- k
def k(G) -> np.ndarray:
    """Node degree = number of non-zeros per row (``src/graphs.jl:9-11``).

    Degree is a *count*, never a weight sum, also for weighted edges.
    Vector in -> scalar count; matrix in -> (N,1) integer column, like ``mapslices``.
    """
    G = np.asarray(G)
    if G.ndim == 1:
        return np.int64(np.count_nonzero(G))
    return np.count_nonzero(G, axis=1).astype(np.int64).reshape(-1, 1)


# --------------------------------------------------------------------------- cutoff / featurize
def cutoff(x, alpha: float, weighted: bool = False):
    """``x >= alpha ? (weighted ? x : 1.0) : 0.0`` element-wise (``src/core.jl:37-43,55-60``).

    ``>=`` is inclusive (``test/runtests.jl:46-47``); the unweighted value is the
    literal 1.0 whatever the input looked like.
    """
    xa = np.asarray(x, dtype=np.float64)
    keep = xa >= alpha
    out = np.where(keep, xa if weighted else 1.0, 0.0)
    return float(out) if out.ndim == 0 else out


def featurize(X: Named, alpha: float, weighted: bool = True) -> Named:
    """Cutoff on the values, columns renamed ``"f" * name`` (``src/core.jl:106-112``)."""
    return Named(cutoff(X.array, alpha, weighted), list(X.rows), ["f" + c for c in X.cols])


# --------------------------------------------------------------------------- construct
def _assert_names_differ(features: Sequence[str], sources: Sequence[str], msg: str) -> None:
    # The reference compares the two *sorted* name vectors element by element
    # (``all(sort(features) .!= sort(sources))``, src/core.jl:156,231,314).  With unequal
    # lengths Julia's broadcast raises DimensionMismatch unless one side has length 1.
    f, s = sorted(features), sorted(sources)
    if len(f) != len(s) and len(f) != 1 and len(s) != 1:
        raise ValueError("DimensionMismatch: arrays could not be broadcast to a common size")
    n = max(len(f), len(s))
    ff = f * n if len(f) == 1 else f
    ss = s * n if len(s) == 1 else s
    if not all(a != b for a, b in zip(ff, ss)):
        raise AssertionError(msg)


def _block_graph(q: List[str], s: List[str], f: List[str], t: List[str],
                 Mqf: np.ndarray, Msf: np.ndarray, Mst: np.ndarray) -> Named:
    nq, ns, nf, nt = len(q), len(s), len(f), len(t)
    n = nq + ns + nf + nt
    A = np.zeros((n, n))
    oq, os_, of, ot = 0, nq, nq + ns, nq + ns + nf
    A[oq:oq + nq, of:of + nf] = Mqf
    A[os_:os_ + ns, of:of + nf] = Msf
    A[os_:os_ + ns, ot:ot + nt] = Mst
    A[of:of + nf, oq:oq + nq] = Mqf.T
    A[of:of + nf, os_:os_ + ns] = Msf.T
    A[ot:ot + nt, os_:os_ + ns] = Mst.T
    names = q + s + f + t
    return Named(A, names, names)


def construct_queries(y: Named, X: Named, queries: Sequence[str]) -> Tuple[Named, Named]:
    """``construct(y, X, queries)`` for k-fold / LOO (``src/core.jl:148-201``).

    Node order is ``[queries; sources; features; targets]`` (pinned by
    ``test/runtests.jl:97-98``).  A feature column is dropped when its name with *all*
    leading ``'f'`` characters stripped is one of the queries (``:152``).
    """
    if y.array.shape[0] != X.array.shape[0]:
        raise AssertionError("Labels and features have different number of source nodes")
    queries = [str(n) for n in queries]
    qset = set(queries)
    features = [f for f in X.cols if f.lstrip("f") not in qset]
    sources = [d for d in X.rows if d not in qset]
    targets = list(y.cols)
    _assert_names_differ(features, sources, "Source and Features nodes have the same names!")
    Mqf = X.sub(queries, features).array
    Msf = X.sub(sources, features).array
    Mst = y.sub(sources, targets).array
    A = _block_graph(queries, sources, features, targets, Mqf, Msf, Mst)
    B = A.copy()
    nq = len(queries)
    B.array[:nq, :] = 0.0
    B.array[:, :nq] = 0.0
    return A, B


def construct_split(ytrain: Named, ytest: Named, Xtrain: Named, Xtest: Named) -> Tuple[Named, Named]:
    """``construct((ytrain,ytest),(Xtrain,Xtest))`` / 4-arg form (``src/core.jl:217-276,294-296``)."""
    if ytrain.array.shape[1] != ytest.array.shape[1]:
        raise AssertionError("Number of targets between test and training sets doesn't match")
    if Xtrain.array.shape[1] != Xtest.array.shape[1]:
        raise AssertionError("Number of features between test and training sets doesn't match")
    features, sources = list(Xtrain.cols), list(ytrain.rows)
    targets, queries = list(ytrain.cols), list(ytest.rows)
    _assert_names_differ(features, sources, "Features and drugs have the same names!")
    A = _block_graph(queries, sources, features, targets, Xtest.array, Xtrain.array, ytrain.array)
    B = A.copy()
    nq = len(queries)
    B.array[:nq, :] = 0.0
    B.array[:, :nq] = 0.0
    return A, B


def construct_single(y: Named, X: Named) -> Named:
    """3-layer ``construct(y, X)``: nodes ``[sources; features; targets]`` (``src/core.jl:308-337``)."""
    features, sources, targets = list(X.cols), list(y.rows), list(y.cols)
    _assert_names_differ(features, sources, "Source and feature nodes have the same names")
    return _block_graph([], sources, features, targets,
                        np.zeros((0, len(features))), X.array, y.array)


# --------------------------------------------------------------------------- spread / predict / clean!
def spread(G) -> np.ndarray:
    """Row-normalise by the non-zero count; ``Inf``/``NaN`` -> 0 (``src/core.jl:365-371``)."""
    G = np.asarray(G, dtype=np.float64)
    deg = k(G).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        W = G / deg
    W[~np.isfinite(W)] = 0.0
    return W


def predict(A: Named, B: Named, ytest: Named) -> Named:
    """``F = A * W^2`` with ``W = spread(B)``; return the block named by ``ytest``
    (``src/core.jl:402-425``).  Any row names present in ``A`` may be asked for
    (query *or* source rows)."""
    W = spread(B.array)
    F = A.array @ (W @ W)
    return Named(F, A.rows, A.cols).sub(ytest.rows, ytest.cols)


def predict_single(A: Named, ytrain: Named) -> Named:
    """``predict(A, ytrain)``: ``W = spread(A)`` on the 3-layer graph (``src/core.jl:446-466``)."""
    W = spread(A.array)
    F = A.array @ (W @ W)
    return Named(F, A.rows, A.cols).sub(ytrain.rows, ytrain.cols)


def clean(yhat: Named, A: Named, y: Named) -> None:
    """``clean!``: column ``t`` of ``yhat`` becomes -99 when target ``t`` has degree 0
    in ``A`` (``src/core.jl:478-484``).  In place."""
    tgt = A.sub(y.cols, A.cols).array
    deg = k(tgt).ravel()
    ci = {n: i for i, n in enumerate(yhat.cols)}
    for t, d in zip(y.cols, deg):
        if d == 0:
            yhat.array[:, ci[t]] = -99.0


# --------------------------------------------------------------------------- factored (sparse) forms
def _inv_count(d: np.ndarray) -> np.ndarray:
    d = np.asarray(d, dtype=np.float64)
    out = np.zeros_like(d)
    nz = d > 0
    out[nz] = 1.0 / d[nz]
    return out


def degrees(Xs, Ys):
    """kf, ks, kt of the query-free graph B."""
    Xs = sp.csr_matrix(Xs)
    Ys = sp.csr_matrix(Ys)
    Xs.eliminate_zeros()
    Ys.eliminate_zeros()
    kf = np.asarray((Xs != 0).sum(axis=0)).ravel().astype(np.int64)
    ks = (np.asarray((Xs != 0).sum(axis=1)).ravel() + np.asarray((Ys != 0).sum(axis=1)).ravel()).astype(np.int64)
    kt = np.asarray((Ys != 0).sum(axis=0)).ravel().astype(np.int64)
    return kf, ks, kt


def predict_factored(Xq, Xs, Ys, rows: str = "query") -> np.ndarray:
    """Factored form of ``predict`` (fp64, scipy.sparse).

    rows="query"  : ``Yq = (Xq D_f^-1) Xs' (D_s^-1 Ys)``                       (Nq x Nt)
    rows="source" : ``(Xs D_f^-1) Xs' (D_s^-1 Ys) + (Ys D_t^-1) Ys' (D_s^-1 Ys)``  (Ns x Nt)
    Equals the corresponding block of the literal ``A @ W @ W`` (checked in tests).
    """
    Xs = sp.csr_matrix(Xs, dtype=np.float64)
    Ys = sp.csr_matrix(Ys, dtype=np.float64)
    kf, ks, kt = degrees(Xs, Ys)
    Df, Ds, Dt = sp.diags(_inv_count(kf)), sp.diags(_inv_count(ks)), sp.diags(_inv_count(kt))
    R = Ds @ Ys
    if rows == "query":
        Xq = sp.csr_matrix(Xq, dtype=np.float64)
        T = (Xq @ Df) @ Xs.T
    elif rows == "source":
        T = (Xs @ Df) @ Xs.T + (Ys @ Dt) @ Ys.T
    else:
        raise ValueError(rows)
    return np.asarray((T @ R).todense())


def transfer_factored(Xq, Xs, Ys, rows: str = "query") -> np.ndarray:
    """The dense transfer block ``T = T0 D_s^-1`` (stage-1 output of the device path)."""
    Xs = sp.csr_matrix(Xs, dtype=np.float64)
    Ys = sp.csr_matrix(Ys, dtype=np.float64)
    kf, ks, kt = degrees(Xs, Ys)
    Df, Ds, Dt = sp.diags(_inv_count(kf)), sp.diags(_inv_count(ks)), sp.diags(_inv_count(kt))
    if rows == "query":
        T = (sp.csr_matrix(Xq, dtype=np.float64) @ Df) @ Xs.T
    else:
        T = (Xs @ Df) @ Xs.T + (Ys @ Dt) @ Ys.T
    return np.asarray((T @ Ds).todense())


def predict_loo_factored(X, Y, clean_flag: bool = False, queries: Iterable[int] | None = None) -> np.ndarray:
    """Leave-one-out scores for every source via the rank-1 degree corrections
    (SURVEY.md section 3.2 "Leave-one-out identity").

    ``X`` is the square featurized similarity (column j is the feature named after
    source j, ``construct(y, X, [i])`` drops column i and row i, src/core.jl:152-153);
    ``Y`` the source x target labels.  Row i of the result is what
    ``predict(construct(y, X, [name_i]), y[[name_i], :])`` returns.
    """
    X = sp.csr_matrix(X, dtype=np.float64)
    Y = sp.csr_matrix(Y, dtype=np.float64)
    X.eliminate_zeros()
    Y.eliminate_zeros()
    n = X.shape[0]
    assert X.shape[1] == n
    Xc = X.tocsc()
    kf = np.asarray((X != 0).sum(axis=0)).ravel().astype(np.float64)
    ks = (np.asarray((X != 0).sum(axis=1)).ravel() + np.asarray((Y != 0).sum(axis=1)).ravel()).astype(np.float64)
    kt = np.asarray((Y != 0).sum(axis=0)).ravel().astype(np.float64)
    qs = list(range(n)) if queries is None else list(queries)
    out = np.zeros((len(qs), Y.shape[1]))
    for o, i in enumerate(qs):
        lo, hi = X.indptr[i], X.indptr[i + 1]
        fidx, fval = X.indices[lo:hi], X.data[lo:hi]
        u = np.zeros(n)
        u[fidx] = fval * _inv_count(kf[fidx] - 1.0)
        u[i] = 0.0
        v = X @ u
        ksi = ks.copy()
        ci = Xc.indices[Xc.indptr[i]:Xc.indptr[i + 1]]
        ksi[ci] -= 1.0
        z = v * _inv_count(ksi)
        z[i] = 0.0
        row = Y.T @ z
        if clean_flag:
            yi = np.zeros(Y.shape[1])
            yi[Y.indices[Y.indptr[i]:Y.indptr[i + 1]]] = 1.0
            row = np.where((kt - yi) == 0, -99.0, row)
        out[o] = row
    return out


def predict_loo_dense(X, Y, clean_flag: bool = False, queries: Iterable[int] | None = None) -> np.ndarray:
    """The same leave-one-out identity as predict_loo_factored for a DENSE featurized similarity ``X`` (a numpy
    array, e.g. cutoff(S, alpha, weighted) of the dense-similarity regime, 90 % full): plain dense mat-vecs, no
    sparse conversion of X.  ``Y`` stays sparse.  tests/test_oracle.py checks it against predict_loo_factored."""
    X = np.asarray(X, dtype=np.float64)
    Y = sp.csr_matrix(Y, dtype=np.float64)
    Y.eliminate_zeros()
    n = X.shape[0]
    assert X.shape == (n, n)
    nzX = X != 0
    kf = nzX.sum(axis=0).astype(np.float64)
    ks = nzX.sum(axis=1).astype(np.float64) + np.asarray((Y != 0).sum(axis=1)).ravel()
    kt = np.asarray((Y != 0).sum(axis=0)).ravel().astype(np.float64)
    qs = list(range(n)) if queries is None else list(queries)
    out = np.zeros((len(qs), Y.shape[1]))
    YT = Y.T.tocsr()
    for o, i in enumerate(qs):
        u = X[i] * _inv_count(kf - nzX[i])     # the query leaves every feature column it touched (src/core.jl:153)
        u[i] = 0.0                             # its own feature column is dropped (src/core.jl:152)
        v = X @ u
        z = v * _inv_count(ks - nzX[:, i])     # feature column f_i is gone from every source row
        z[i] = 0.0
        row = YT @ z
        if clean_flag:
            yi = np.zeros(Y.shape[1])
            yi[Y.indices[Y.indptr[i]:Y.indptr[i + 1]]] = 1.0
            row = np.where((kt - yi) == 0, -99.0, row)
        out[o] = row
    return out


# --------------------------------------------------------------------------- synthetic inputs shared by tests / bench
def synth_bipartite(nq: int, ns: int, nf: int, nt: int, dx: float, dy: float, seed: int,
                    weighted: bool = True, alpha: float = 0.5, dtype=np.float32):
    """Seeded synthetic inputs of the BASELINE config shapes.

    Returns scipy CSR ``Xq (nq x nf)``, ``Xs (ns x nf)``, ``Ys (ns x nt)`` with sorted
    indices.  Values of X are U(alpha, 1] when weighted else 1; Y is {0,1}.  When
    ns == nf the diagonal of Xs is forced non-zero (self-similarity).
    """
    rng = np.random.default_rng(seed)

    def rand_csr(r, c, d, vals):
        m = sp.random(r, c, density=d, format="csr", random_state=rng, dtype=np.float64)
        m.data = vals(m.nnz)
        m.sort_indices()
        return m

    xv = (lambda n: alpha + (1.0 - alpha) * (1.0 - rng.random(n))) if weighted else (lambda n: np.ones(n))
    Xq = rand_csr(nq, nf, dx, xv)
    Xs = rand_csr(ns, nf, dx, xv)
    if ns == nf:
        Xs = Xs.tolil()
        Xs.setdiag(1.0)
        Xs = Xs.tocsr()
        Xs.sort_indices()
    Ys = rand_csr(ns, nt, dy, lambda n: np.ones(n))
    return Xq.astype(dtype), Xs.astype(dtype), Ys.astype(dtype)


# --------------------------------------------------------------------------- threshold-free metrics
# Literal restatement of src/performance.jl:22-89 with MLBase's roc(gt, scores, thresholds) (positive when
# score >= threshold) and Trapz.trapz.  The reference's own tests for these are `skip = true`
# (test/runtests.jl:210-223): parity unpinned.
def _confusion_rates(y, yhat):
    y = np.asarray(y).ravel() != 0
    yhat = np.asarray(yhat, dtype=np.float64).ravel()
    thresholds = np.unique(yhat)                      # sort(unique(yhat))
    P, N = y.sum(), (~y).sum()
    tp = np.array([(y & (yhat >= t)).sum() for t in thresholds], dtype=np.float64)
    fp = np.array([(~y & (yhat >= t)).sum() for t in thresholds], dtype=np.float64)
    return tp, fp, float(P), float(N)


def _trapz(x, y):
    return float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / 2.0))


def auroc(y, yhat):
    """src/performance.jl:49-63"""
    tp, fp, P, N = _confusion_rates(y, yhat)
    with np.errstate(invalid="ignore", divide="ignore"):
        return abs(_trapz(fp / N, tp / P))


def auprc(y, yhat):
    """src/performance.jl:74-89"""
    tp, fp, P, N = _confusion_rates(y, yhat)
    with np.errstate(invalid="ignore", divide="ignore"):
        return abs(_trapz(tp / P, tp / (tp + fp)))


def bedroc(y, yhat, rev=True, alpha=20.0):
    """src/performance.jl:22-38; sortperm is stable, so ties keep their position order"""
    y = np.asarray(y).ravel() != 0
    yhat = np.asarray(yhat, dtype=np.float64).ravel()
    N, n = len(y), int(y.sum())
    order = np.argsort(-yhat if rev else yhat, kind="stable")
    r = np.flatnonzero(y[order]) + 1
    s = np.sum(np.exp(-alpha * r / N))
    Ra = n / N
    rand_sum = Ra * (1 - np.exp(-alpha)) / (np.exp(alpha / N) - 1)
    fac = Ra * np.sinh(alpha / 2) / (np.cosh(alpha / 2) - np.cosh(alpha / 2 - alpha * Ra))
    cte = 1 / (1 - np.exp(alpha * (1 - Ra)))
    return float(s * fac / rand_sum + cte)


def validity_ratio(yhat):
    """src/performance.jl:558-560"""
    yhat = np.asarray(yhat).ravel()
    return float(np.count_nonzero(yhat) / len(yhat))
