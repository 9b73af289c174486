"""Host-side mirror of SimSpread.jl's hot-path API (src/SimSpread.jl:21-56), same names, argument
meaning and error behaviour, computing on the MI355X through libsimspread_hip.so.

    k, cutoff, featurize, construct (4 methods), spread, predict (3 methods), clean (= clean!)

Julia is not installed where this is built, so this mirror is Python; the Julia binding a
maintainer would drop into the reference is julia/SimSpreadHIP.jl.

Differences that are deliberate and visible:
  * ``construct`` returns light ``Network`` objects that keep the blocks Xq/Xs/Ys instead of
    materialising the dense N x N matrices; ``Network.array`` builds the dense matrix on demand
    (small graphs only) so code that inspects ``A.array`` / ``names(A, 1)`` keeps working.
  * ``GPU=`` keyword: accepted and ignored -- this build always runs on the GPU (fp64 by
    default, like the reference's CPU path; ``precision="f32"`` selects what ``GPU=true`` used).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib as L
from .engine import DeviceGraph


# --------------------------------------------------------------------------- NamedMatrix
class NamedMatrix:
    """Matrix + row/column names (stand-in for NamedArrays.NamedMatrix used throughout the reference)."""

    def __init__(self, array, rows: Optional[Sequence] = None, cols: Optional[Sequence] = None):
        # NamedArray([1 0; 0 1]) keeps Int elements in Julia (they print bare in save); remember that
        self.integer = np.issubdtype(np.asarray(array).dtype, np.integer)
        self.array = np.array(array, dtype=np.float64, ndmin=2)
        r, c = self.array.shape
        # NamedArrays' default names are "1".."n"
        self.rows = [str(x) for x in (rows if rows is not None else range(1, r + 1))]
        self.cols = [str(x) for x in (cols if cols is not None else range(1, c + 1))]
        if len(self.rows) != r or len(self.cols) != c:
            raise ValueError("name lists do not match matrix shape")
        if len(set(self.rows)) != r or len(set(self.cols)) != c:
            raise ValueError("duplicate names")

    def names(self, dim: int) -> List[str]:
        return list(self.rows if dim == 1 else self.cols)

    def setnames(self, names: Sequence, dim: int) -> None:
        names = [str(n) for n in names]
        if dim == 1:
            assert len(names) == self.array.shape[0]
            self.rows = names
        else:
            assert len(names) == self.array.shape[1]
            self.cols = names

    @property
    def shape(self):
        return self.array.shape

    def sub(self, rows: Sequence, cols: Sequence) -> "NamedMatrix":
        ri = {n: i for i, n in enumerate(self.rows)}
        ci = {n: i for i, n in enumerate(self.cols)}
        try:
            r = [ri[str(n)] for n in rows]
            c = [ci[str(n)] for n in cols]
        except KeyError as e:
            raise KeyError(f"name {e} not found") from None
        out = NamedMatrix(self.array[np.ix_(r, c)], [str(n) for n in rows], [str(n) for n in cols])
        out.integer = self.integer          # indexing an Int NamedArray gives an Int NamedArray in Julia
        return out

    def copy(self) -> "NamedMatrix":
        out = NamedMatrix(self.array.copy(), list(self.rows), list(self.cols))
        out.integer = self.integer
        return out

    def __eq__(self, other):  # NamedArrays `==` compares values only
        if isinstance(other, NamedMatrix):
            return self.array.shape == other.array.shape and bool((self.array == other.array).all())
        return NotImplemented

    def __repr__(self):
        return f"NamedMatrix({self.array.shape[0]}x{self.array.shape[1]})"


def names(M, dim: int) -> List[str]:
    return M.names(dim)


# --------------------------------------------------------------------------- k / cutoff / featurize / spread
def _dev_dtype(precision: str):
    return {"f64": np.float64, "f32": np.float32}[precision]


def k(G, *args):
    """Node degrees = number of non-zeros per row (src/graphs.jl:9-11).

    k(v) -> count for a vector; k(G) -> (N,1) integer matrix; k(i, G) -> count of (1-based) row i.
    """
    if args:  # k(i, G)
        i, G = G, args[0]
        return k(np.asarray(G.array if isinstance(G, NamedMatrix) else G)[int(i) - 1, :])
    a = np.asarray(G.array if isinstance(G, NamedMatrix) else G)
    if a.ndim == 1:
        a = a.reshape(1, -1)
        vec = True
    else:
        vec = False
    a = np.asfortranarray(a, dtype=np.float64)
    deg = np.zeros(a.shape[0], np.int64)
    L.check(L.lib().ss_row_degree_f64(a.ctypes.data, a.shape[0], a.shape[1], max(a.shape[0], 1), deg.ctypes.data,
                                      L.SS_MEM_HOST))
    return np.int64(deg[0]) if vec else deg.reshape(-1, 1)


def cutoff(x, alpha: float, weighted: bool = False):
    """x >= alpha ? (weighted ? x : 1.0) : 0.0 (src/core.jl:37-43,55-60)."""
    if np.ndim(x) == 0:
        a = np.array([[x]], dtype=np.float64)
        scalar = True
    else:
        a = np.asarray(x, dtype=np.float64)
        scalar = False
    shape = a.shape
    a2 = np.asfortranarray(a.reshape(shape[0], -1))
    out = np.empty_like(a2, order="F")
    L.check(L.lib().ss_cutoff_f64(a2.ctypes.data, a2.shape[0], a2.shape[1], max(a2.shape[0], 1), float(alpha),
                                  1 if weighted else 0, out.ctypes.data, max(a2.shape[0], 1), L.SS_MEM_HOST))
    return float(out[0, 0]) if scalar else np.ascontiguousarray(out).reshape(shape)


def featurize(X: NamedMatrix, alpha: float, weighted: bool = True) -> NamedMatrix:
    """Similarity cutoff + column rename "f" * name (src/core.jl:106-112)."""
    out = NamedMatrix(cutoff(X.array, alpha, weighted), list(X.rows), ["f" + c for c in X.cols])
    return out


def spread(G):
    """Transfer matrix W = G ./ k(G), zero-degree rows -> 0 (src/core.jl:365-371,373,375-380)."""
    named = isinstance(G, NamedMatrix)
    a = np.asfortranarray(np.asarray(G.array if named else G, dtype=np.float64))
    out = np.empty_like(a, order="F")
    L.check(L.lib().ss_spread_f64(a.ctypes.data, a.shape[0], a.shape[1], max(a.shape[0], 1), out.ctypes.data,
                                  max(a.shape[0], 1), L.SS_MEM_HOST))
    out = np.ascontiguousarray(out)
    return NamedMatrix(out, G.rows, G.cols) if named else out


# --------------------------------------------------------------------------- construct
class Network:
    """What ``construct`` returns in place of a dense named N x N adjacency matrix.

    ``kind`` is "A" (full graph), "B" (query rows/columns zeroed, src/core.jl:196-198) or
    "single" (3-layer graph of src/core.jl:308-337).  Node order is [queries; sources; features;
    targets] (test/runtests.jl:97-98)."""

    def __init__(self, kind, queries, sources, features, targets, Xq, Xs, Ys):
        self.kind = kind
        self.queries, self.sources, self.features, self.targets = queries, sources, features, targets
        self.Xq, self.Xs, self.Ys = Xq, Xs, Ys
        self._names = list(queries) + list(sources) + list(features) + list(targets)
        if len(set(self._names)) != len(self._names):
            raise ValueError("duplicate node names in the network")
        self._dev = {}

    def names(self, dim: int) -> List[str]:
        return list(self._names)

    @property
    def rows(self):
        return list(self._names)

    @property
    def cols(self):
        return list(self._names)

    @property
    def array(self) -> np.ndarray:
        nq, ns, nf, nt = len(self.queries), len(self.sources), len(self.features), len(self.targets)
        n = nq + ns + nf + nt
        A = np.zeros((n, n))
        oq, os_, of, ot = 0, nq, nq + ns, nq + ns + nf
        if self.kind != "B" and nq:
            A[oq:oq + nq, of:of + nf] = self.Xq
            A[of:of + nf, oq:oq + nq] = self.Xq.T
        A[os_:os_ + ns, of:of + nf] = self.Xs
        A[os_:os_ + ns, ot:ot + nt] = self.Ys
        A[of:of + nf, os_:os_ + ns] = self.Xs.T
        A[ot:ot + nt, os_:os_ + ns] = self.Ys.T
        return A

    def device(self, precision: str = "f64") -> DeviceGraph:
        if precision not in self._dev:
            self._dev[precision] = DeviceGraph.from_dense(self.Xq if len(self.queries) else None, self.Xs, self.Ys,
                                                          alpha=None, dtype=_dev_dtype(precision))
        return self._dev[precision]


def _assert_names_differ(features, sources, msg):
    # the reference compares the two sorted name vectors element-wise (src/core.jl:156,231,314):
    # unequal lengths raise DimensionMismatch (unless one side has one element)
    f, s = sorted(features), sorted(sources)
    if len(f) != len(s) and len(f) != 1 and len(s) != 1:
        raise ValueError("DimensionMismatch: arrays could not be broadcast to a common size; "
                         f"got a dimension with lengths {len(f)} and {len(s)}")
    n = max(len(f), len(s))
    ff = f * n if len(f) == 1 else f
    ss = s * n if len(s) == 1 else s
    if not all(a != b for a, b in zip(ff, ss)):
        raise AssertionError(msg)


def construct(*args):
    """construct(y, X, queries) | construct((ytrain,ytest),(Xtrain,Xtest)) |
    construct(ytrain, ytest, Xtrain, Xtest) | construct(y, X)   (src/core.jl:148-201,217-276,294-296,308-337)."""
    if len(args) == 3:
        y, X, queries = args
        if y.shape[0] != X.shape[0]:
            raise AssertionError("Labels and features have different number of source nodes")
        queries = [str(q) for q in queries]
        qset = set(queries)
        features = [f for f in X.cols if f.lstrip("f") not in qset]
        sources = [d for d in X.rows if d not in qset]
        targets = list(y.cols)
        _assert_names_differ(features, sources, "Source and Features nodes have the same names!")
        Xq = X.sub(queries, features).array
        Xs = X.sub(sources, features).array
        Ys = y.sub(sources, targets).array
        A = Network("A", queries, sources, features, targets, Xq, Xs, Ys)
        B = Network("B", queries, sources, features, targets, Xq, Xs, Ys)
        B._dev = A._dev
        return A, B
    if len(args) == 4:
        ytrain, ytest, Xtrain, Xtest = args
        return construct((ytrain, ytest), (Xtrain, Xtest))
    if len(args) == 2 and isinstance(args[0], tuple):
        (ytrain, ytest), (Xtrain, Xtest) = args
        if ytrain.shape[1] != ytest.shape[1]:
            raise AssertionError("Number of targets between test and training sets doesn't match")
        if Xtrain.shape[1] != Xtest.shape[1]:
            raise AssertionError("Number of features between test and training sets doesn't match")
        features, sources = list(Xtrain.cols), list(ytrain.rows)
        targets, queries = list(ytrain.cols), list(ytest.rows)
        _assert_names_differ(features, sources, "Features and drugs have the same names!")
        A = Network("A", queries, sources, features, targets, Xtest.array, Xtrain.array, ytrain.array)
        B = Network("B", queries, sources, features, targets, Xtest.array, Xtrain.array, ytrain.array)
        B._dev = A._dev
        return A, B
    if len(args) == 2:
        y, X = args
        features, sources, targets = list(X.cols), list(y.rows), list(y.cols)
        _assert_names_differ(features, sources, "Source and feature nodes have the same names")
        return Network("single", [], sources, features, targets, np.zeros((0, len(features))), X.array, y.array)
    raise TypeError("construct: no method matching these arguments")


# --------------------------------------------------------------------------- predict / clean!
def _predict_network(A: Network, y, precision: str) -> NamedMatrix:
    dev = A.device(precision)
    qi = {n: i for i, n in enumerate(A.queries)}
    si = {n: i for i, n in enumerate(A.sources)}
    ti = {n: i for i, n in enumerate(A.targets)}
    rows, cols = y.names(1), y.names(2)
    if not all(c in ti for c in cols) or not all((r in qi) or (r in si) for r in rows):
        return None  # asks for a block outside [queries|sources] x targets: general path
    out = np.zeros((len(rows), len(cols)))
    tcols = [ti[c] for c in cols]
    q_rows = [(o, qi[r]) for o, r in enumerate(rows) if r in qi]
    s_rows = [(o, si[r]) for o, r in enumerate(rows) if r not in qi]
    if q_rows:
        lo, hi = min(i for _, i in q_rows), max(i for _, i in q_rows) + 1
        blk = dev.predict("query", lo, hi)
        out[[o for o, _ in q_rows]] = blk[[i - lo for _, i in q_rows]][:, tcols]
    if s_rows:
        lo, hi = min(i for _, i in s_rows), max(i for _, i in s_rows) + 1
        blk = dev.predict("source", lo, hi)
        out[[o for o, _ in s_rows]] = blk[[i - lo for _, i in s_rows]][:, tcols]
    return NamedMatrix(out, rows, cols)


def _predict_general(A, B, y, precision: str) -> NamedMatrix:
    import scipy.sparse as sp
    An, Bn = A.names(1), B.names(1)
    Aa, Ba = np.asarray(A.array, dtype=np.float64), np.asarray(B.array, dtype=np.float64)
    if Aa.shape != Ba.shape or Aa.shape[0] != Aa.shape[1]:
        raise ValueError("DimensionMismatch: A and B must be square matrices of the same size")
    ri = {n: i for i, n in enumerate(An)}
    ci = {n: i for i, n in enumerate(A.names(2))}
    rows, cols = y.names(1), y.names(2)
    r = [ri[n] for n in rows]
    c = [ci[n] for n in cols]
    dev = DeviceGraph.general(sp.csr_matrix(Aa[r, :]), sp.csr_matrix(Ba), sp.csr_matrix(Ba[:, c].T),
                              dtype=_dev_dtype(precision))
    out = dev.predict("query")
    dev.close()
    return NamedMatrix(np.asarray(out, dtype=np.float64), rows, cols)


def predict(*args, GPU: bool = False, precision: str = "f64") -> NamedMatrix:
    """predict((A,B), ytest) | predict(A, B, ytest) | predict(A, ytrain)   (src/core.jl:402-425,446-466).

    Returns the block of F = A * spread(B)^2 named by the rows/columns of the last argument, as a
    NamedMatrix of Float64 (the reference widens GPU results back to Float64, src/core.jl:413)."""
    if len(args) == 3:
        A, B, y = args
    elif len(args) == 2 and isinstance(args[0], tuple):
        (A, B), y = args
    elif len(args) == 2:
        A, y = args
        B = A  # predict(A, ytrain): W = spread(A)
    else:
        raise TypeError("predict: no method matching these arguments")
    if GPU and precision == "f64":
        precision = "f32"  # what the reference's GPU=true computed in (src/core.jl:404)
    network_pair = (isinstance(A, Network) and isinstance(B, Network) and B._dev is A._dev and
                    (B.kind == "B" or (B is A and not A.queries)))
    if network_pair:
        res = _predict_network(A, y, precision)
        if res is not None:
            return res
    return _predict_general(A, B, y, precision)


def clean(yhat: NamedMatrix, A, y) -> None:
    """clean!(yhat, A, y): column t of yhat becomes -99 when target t has degree 0 in A (src/core.jl:478-484)."""
    tnames = y.names(2)
    if isinstance(A, Network):
        ti = {n: i for i, n in enumerate(A.targets)}
        if all(t in ti for t in tnames):
            _, _, kt = A.device("f64").degrees()
            deg = [kt[ti[t]] for t in tnames]
        else:
            deg = k(_rows_of(A, tnames)).ravel()
    else:
        deg = k(_rows_of(A, tnames)).ravel()
    ci = {n: i for i, n in enumerate(yhat.cols)}
    for t, d in zip(tnames, deg):
        if d == 0:
            yhat.array[:, ci[t]] = -99.0


def _rows_of(A, rownames):
    idx = {n: i for i, n in enumerate(A.names(1))}
    return np.asarray(A.array)[[idx[n] for n in rownames], :]


clean_ = clean  # Julia's `clean!`


# --------------------------------------------------------------------------- split / save ("next" rows of SURVEY 8f)
def split(y: NamedMatrix, k: int, seed: int = 1) -> List[List[str]]:
    """k-fold grouping of the source names: ``shuffle!(MersenneTwister(seed), sources)``, then source i (1-based, in
    shuffled order) goes to fold mod(i, k) + 1 (src/core.jl:11-25).  The shuffle is Julia's, bit for bit
    (julia_rng.py: dSFMT-19937 + Random.shuffle!), so the groups are the reference's for the same seed -- pinned by the
    reference's own expected grouping for seed 1, k 5 (test/runtests.jl:31-32)."""
    from .julia_rng import shuffle
    sources = shuffle(list(y.names(1)), seed)
    groups: List[List[str]] = [[] for _ in range(k)]
    for i, s in enumerate(sources, start=1):
        groups[i % k].append(s)
    return groups


def _julia_number(x) -> str:
    """How Julia's join/print shows a matrix element: Ints bare, floats shortest round-trip with a trailing .0"""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    f = float(x)
    if f == int(f) and abs(f) < 1e16:
        return f"{int(f)}.0" if not np.isnan(f) else "NaN"
    return repr(f)


def save(filepath: str, *args, delimiter: str = "\t") -> None:
    """save(filepath, yhat, y; delimiter) | save(filepath, fidx, yhat, y; delimiter)   (src/core.jl:503-522,542-561).

    Appends one line per (query, target): fold, "query", "target", score, label.  Without a fold index the
    first column is the 1-based position of the query.  Integer-valued NamedMatrix payloads built from integer
    input print bare (test/data/save1-4); this mirror keeps that by remembering the input dtype."""
    if len(args) == 2:
        fidx, (yhat, y) = None, args
    elif len(args) == 3:
        fidx, yhat, y = args
    else:
        raise TypeError("save: no method matching these arguments")
    queries, targets = y.names(1), y.names(2)
    qi = {n: i for i, n in enumerate(yhat.rows)}
    ti = {n: i for i, n in enumerate(yhat.cols)}
    yq = {n: i for i, n in enumerate(y.rows)}
    yt = {n: i for i, n in enumerate(y.cols)}

    def show(M, v):
        return _julia_number(int(v)) if getattr(M, "integer", False) else _julia_number(v)

    with open(filepath, "a+") as f:
        for pos, q in enumerate(queries, start=1):
            for t in targets:
                row = [str(pos if fidx is None else fidx), f'"{q}"', f'"{t}"',
                       show(yhat, yhat.array[qi[q], ti[t]]), show(y, y.array[yq[q], yt[t]])]
                f.write(delimiter.join(row) + "\n")


def save_loo(filepath: str, y: NamedMatrix, X: NamedMatrix, delimiter: str = "\t", block: int = 1024,
             precision: str = "f64", clean_scores: bool = True, graph: Optional[DeviceGraph] = None) -> int:
    """The reference's leave-one-out loop with its output step, streamed from device score blocks:

        for (i, s) in enumerate(names(y, 1))                       # user loop over construct's fold form
            A, B = construct(y, X, [s]); yhat = predict((A, B), y[[s], :]); clean!(yhat, A, y[[s], :])
            save(filepath, i, yhat, y[[s], :]; delimiter)           # src/core.jl:542-561

    All folds come from ONE resident graph (ss_predict_loo_*); `block` folds are scored at a time, copied to the
    host and appended as text in exactly save's wire format (fold, "source", "target", score, label), so the
    full score matrix never exists on the host (BASELINE configs[2]: 40 GB) and an interrupted run can be resumed
    at a block boundary (the file is opened in append mode, like the reference's).  Returns the number of lines."""
    if y.shape[0] != X.shape[0]:
        raise AssertionError("Labels and features have different number of source nodes")
    g = graph if graph is not None else DeviceGraph.from_dense(None, X.array, y.array, alpha=None,
                                                               dtype=_dev_dtype(precision))
    sources, targets = y.names(1), y.names(2)
    ns = len(sources)
    quoted_t = [f'"{t}"' for t in targets]
    show_y = (lambda v: _julia_number(int(v))) if getattr(y, "integer", False) else _julia_number
    lines = 0
    with open(filepath, "a+") as f:
        for lo in range(0, ns, block):
            hi = min(ns, lo + block)
            scores = np.asarray(g.predict_loo(lo, hi, clean=clean_scores), dtype=np.float64)   # one device -> host copy per block
            for r in range(hi - lo):
                i = lo + r
                head = f'{i + 1}{delimiter}"{sources[i]}"{delimiter}'
                yrow, srow = y.array[i], scores[r]
                f.write("".join(head + quoted_t[c] + delimiter + _julia_number(srow[c]) + delimiter + show_y(yrow[c]) + "\n"
                                for c in range(len(targets))))
                lines += len(targets)
    if graph is None:
        g.close()
    return lines


# --------------------------------------------------------------------------- text I/O
def _parse_matrix(M: List[List[str]], rows: bool = False, cols: bool = False, type=float) -> NamedMatrix:
    """Cells of a delimited file -> NamedMatrix (src/utils.jl:26-40): names from the first column / row when
    present, otherwise "R#i" / "C#j" (1-based); rows and columns are then reordered by *string* sort of the names
    (so "R#10" sorts before "R#2", as in the reference)."""
    r0, c0 = (1 if cols else 0), (1 if rows else 0)
    body = [line[c0:] for line in M[r0:]]
    width = len(body[0]) if body else 0
    if any(len(line) != width for line in body):
        raise ValueError("read_namedmatrix: ragged rows")
    values = np.array([[type(v) for v in line] for line in body]).reshape(len(body), width)
    row_names = [str(line[0]) for line in M[r0:]] if rows else [f"R#{i}" for i in range(1, len(body) + 1)]
    col_names = [str(c) for c in M[0][c0:]] if cols else [f"C#{j}" for j in range(1, width + 1)]
    named = NamedMatrix(values, row_names, col_names)
    return named.sub(sorted(row_names), sorted(col_names))


def read_namedmatrix(filepath: str, delimiter: str = " ", valuetype=float, rows: bool = True,
                     cols: bool = True) -> NamedMatrix:
    """read_namedmatrix(filepath, delimiter=' ', valuetype=Float64; rows=true, cols=true)  (src/utils.jl:52-55).
    Every cell is read as a string (readdlm(..., String)) and split on the single delimiter character, so a
    header line that starts with the delimiter has an empty corner cell (test/data/mat1)."""
    with open(filepath) as f:
        M = [line.rstrip("\n").rstrip("\r").split(delimiter) for line in f if line.strip("\r\n") != ""]
    return _parse_matrix(M, rows, cols, type=valuetype)


def writedlm(io, x: NamedMatrix, delimiter: str = "\t") -> None:
    """writedlm(io, x::NamedMatrix[, delimiter])  (src/utils.jl:6-11): an empty corner cell and the column names,
    then one line per row: name, values."""
    own = isinstance(io, str)
    f = open(io, "w") if own else io
    try:
        f.write(delimiter.join([""] + list(x.cols)) + "\n")
        for name, line in zip(x.rows, x.array):
            f.write(delimiter.join([str(name)] + [_julia_number(v) for v in line]) + "\n")
    finally:
        if own:
            f.close()


# --------------------------------------------------------------------------- ranked metrics
def _per_group_hits(y, yhat, grouping, L: int):
    from .engine import topl
    y = np.asarray(y, dtype=np.float64).ravel()
    yhat = np.asarray(yhat, dtype=np.float32).ravel()
    grouping = np.asarray(grouping).ravel()
    if not (len(y) == len(yhat) == len(grouping)):
        raise AssertionError("Number of predictions must match number of labels")
    if L <= 0:
        raise AssertionError("Please use a list length greater than 0 (L > 0)")
    out = []
    for gname in dict.fromkeys(grouping.tolist()):       # unique(), first-seen order
        sel = grouping == gname
        yg, sg = y[sel], yhat[sel]
        if len(yg) <= L:
            raise AssertionError("Number of labels is less than length (L > y)")
        idx, _ = topl(sg.reshape(1, -1), L)              # the L best of the group, on the device
        out.append((yg[idx[0]].sum(), yg.sum()))
    return out


def recallatL(y, yhat, grouping, L: int = 20) -> float:
    """Mean recall@L per group (src/performance.jl:308-352); groups without positives give NaN (as the reference)."""
    vals = [h / t if t > 0 else np.nan for h, t in _per_group_hits(y, yhat, grouping, L)]
    return float(np.mean(vals))


def precisionatL(y, yhat, grouping, L: int = 20) -> float:
    """Mean precision@L per group (src/performance.jl:354-400)."""
    return float(np.mean([h / L for h, _ in _per_group_hits(y, yhat, grouping, L)]))
