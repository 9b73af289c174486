"""Device handles over the C ABI: the tri-partite graph (`DeviceGraph`) and the raw W*R
operand (`DeviceSpMat`).  Inputs are numpy / scipy.sparse (host) or torch CUDA tensors
(device, passed by ``data_ptr()``); outputs are numpy arrays or, when ``out`` is a torch CUDA
tensor, written in place on the device.  Everything computes on the GPU through
libsimspread_hip.so -- nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L
from . import _lib as L_


def _suffix(dtype) -> str:
    dt = np.dtype(dtype)
    if dt == np.float32:
        return "f32"
    if dt == np.float64:
        return "f64"
    raise TypeError(f"dtype {dt} not supported (float32 / float64)")


def _is_torch(x) -> bool:
    if type(x).__module__.startswith("torch"):
        L.use_torch_stream()  # order the library's kernels after the ones that produced this tensor
        return True
    return False


def _csr_parts(m, dtype, shape=None):
    """(ptr int64, idx int32, val dtype) numpy arrays of a scipy.sparse matrix, sorted indices."""
    import scipy.sparse as sp
    if m is None:
        rows = shape[0]
        return np.zeros(rows + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, dtype)
    m = sp.csr_matrix(m)
    if shape is not None and m.shape != tuple(shape):
        raise ValueError(f"block has shape {m.shape}, expected {tuple(shape)}")
    if not m.has_sorted_indices:
        m = m.sorted_indices()
    m.sum_duplicates()
    return (np.ascontiguousarray(m.indptr, dtype=np.int64), np.ascontiguousarray(m.indices, dtype=np.int32),
            np.ascontiguousarray(m.data, dtype=dtype))


def _ptr(a):
    return None if a is None else a.ctypes.data


class DeviceGraph:
    """``construct(...)`` on the device: blocks Xq (nq x nf), Xs (ns x nf), Ys (ns x nt) as CSR plus
    transposes and count degrees (reference: src/core.jl:148-201,217-276,308-337,365-371)."""

    def __init__(self, handle, dtype, general=False):
        self._h = handle
        self.dtype = np.dtype(dtype)
        self._suf = _suffix(dtype)
        self.general = general
        info = (C.c_int64 * 7)()
        L.check(L.load().ss_graph_info(self._h, info))
        self.nq, self.ns, self.nf, self.nt, self.nnz_xq, self.nnz_xs, self.nnz_ys = [int(v) for v in info]

    # ---------------------------------------------------------------- constructors
    @classmethod
    def from_sparse(cls, Xq, Xs, Ys, dtype=np.float32):
        """Host CSR blocks (scipy.sparse or anything csr_matrix accepts).  Xq may be None (3-layer graph)."""
        import scipy.sparse as sp
        lib = L.lib()
        Xs = sp.csr_matrix(Xs)
        Ys = sp.csr_matrix(Ys)
        ns, nf = Xs.shape
        nt = Ys.shape[1]
        if Ys.shape[0] != ns:
            raise AssertionError("Labels and features have different number of source nodes")
        nq = 0 if Xq is None else sp.csr_matrix(Xq).shape[0]
        if Xq is not None and sp.csr_matrix(Xq).shape[1] != nf:
            raise AssertionError("Number of features between test and training sets doesn't match")
        q = _csr_parts(Xq, dtype, (nq, nf))
        s = _csr_parts(Xs, dtype, (ns, nf))
        y = _csr_parts(Ys, dtype, (ns, nt))
        h = C.c_void_p()
        fn = getattr(lib, f"ss_graph_create_csr_{_suffix(dtype)}")
        L.check(fn(nq, ns, nf, nt, _ptr(q[0]), _ptr(q[1]), _ptr(q[2]), _ptr(s[0]), _ptr(s[1]), _ptr(s[2]),
                   _ptr(y[0]), _ptr(y[1]), _ptr(y[2]), 0, L.SS_MEM_HOST, C.byref(h)))
        return cls(h, dtype)

    @classmethod
    def from_device_csr(cls, nq, ns, nf, nt, xq, xs, ys, dtype=np.float32):
        """CSR blocks that already live on the GPU: each of xq, xs, ys is a (ptr int64, idx int32, val)
        triple of torch CUDA tensors (xq may be None; a val of None means all ones)."""
        lib = L.lib()
        import torch

        def parts(t, rows):
            if t is None:
                z = torch.zeros(rows + 1, dtype=torch.int64, device="cuda")
                return z, None, None, (z,)
            ptr, idx, val = t
            _is_torch(ptr)
            want = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
            if ptr.dtype != torch.int64 or idx.dtype != torch.int32 or (val is not None and val.dtype != want):
                raise TypeError("device CSR needs int64 pointers, int32 indices and values of the graph precision")
            return ptr, idx, val, (ptr, idx, val)

        dp = lambda t: None if t is None else t.data_ptr()
        q = parts(xq, nq)
        s_ = parts(xs, ns)
        y = parts(ys, ns)
        h = C.c_void_p()
        fn = getattr(lib, f"ss_graph_create_csr_{_suffix(dtype)}")
        L.check(fn(nq, ns, nf, nt, dp(q[0]), dp(q[1]), dp(q[2]), dp(s_[0]), dp(s_[1]), dp(s_[2]),
                   dp(y[0]), dp(y[1]), dp(y[2]), 0, L.SS_MEM_DEVICE, C.byref(h)))
        return cls(h, dtype)

    @classmethod
    def from_dense(cls, Sq, Ss, Y, alpha: Optional[float] = None, weighted: bool = True, dtype=np.float32):
        """Dense blocks; with ``alpha`` the featurize cutoff (src/core.jl:106-112) is applied on the
        device while the CSR operands are assembled.  Arrays may be numpy (any order) or torch CUDA
        tensors; they are read as column-major (a C-order array is passed as its transpose... no copy
        is made for Fortran-order inputs)."""
        lib = L.lib()
        dt = np.dtype(dtype)

        def prep(a, rows_expected=None):
            # returns (pointer, ld, rows, cols, mem, keepalive) for a column-major view
            if a is None:
                return None, 1, 0, None, L.SS_MEM_HOST, None
            if _is_torch(a):
                import torch
                want = torch.float32 if dt == np.float32 else torch.float64
                t = a.to(want)
                # column-major rows x cols == row-major cols x rows: need t.T contiguous
                tt = t.t().contiguous()
                return tt.data_ptr(), t.shape[0], t.shape[0], t.shape[1], L.SS_MEM_DEVICE, tt
            arr = np.asfortranarray(np.asarray(a, dtype=dt))
            if arr.ndim != 2:
                raise ValueError("dense blocks must be matrices")
            return arr.ctypes.data, max(arr.shape[0], 1), arr.shape[0], arr.shape[1], L.SS_MEM_HOST, arr

        pq, ldq, nq, nfq, memq, kq = prep(Sq)
        ps, lds, ns, nf, mems, ks_ = prep(Ss)
        py, ldy, nsy, nt, memy, ky = prep(Y)
        if nsy != ns:
            raise AssertionError("Labels and features have different number of source nodes")
        if Sq is not None and nfq != nf:
            raise AssertionError("Number of features between test and training sets doesn't match")
        mems_used = {m for m, a in ((memq, Sq), (mems, Ss), (memy, Y)) if a is not None}
        if len(mems_used) != 1:
            raise ValueError("dense blocks must all live on the host or all on the device")
        mem = mems_used.pop()
        h = C.c_void_p()
        fn = getattr(lib, f"ss_graph_create_dense_{_suffix(dtype)}")
        ctype = C.c_float if dt == np.float32 else C.c_double
        L.check(fn(nq, ns, nf, nt, pq, ldq, ps, lds, py, ldy, 0 if alpha is None else 1,
                   ctype(0.0 if alpha is None else alpha), 1 if weighted else 0, mem, C.byref(h)))
        del kq, ks_, ky
        return cls(h, dtype)

    @classmethod
    def from_similarity(cls, Sq, Ss, Y, alpha: float, weighted: bool = True, dtype=np.float32):
        """Dense-similarity regime: raw similarities Sq (nq x ns, may be None) and Ss (ns x ns) stay dense on the
        device, the cutoff is applied inside the MFMA stage-1 product; Y (ns x nt) is sparse.  dtype float32: bf16
        matrix cores on exact bf16 planes (the reference's GPU=true precision); float64: the fp64 matrix instruction
        (the reference's default precision).  Inputs: numpy arrays / scipy matrix on the host, or torch CUDA tensors
        for Sq, Ss with Y = (ptr, idx, val) device CSR.  Serves predict (query / source rows), predict_loo and
        predict_kfold."""
        import scipy.sparse as sp
        lib = L.lib()
        dev = _is_torch(Ss)
        keep = []
        dt = np.dtype(dtype).type
        if dt not in (np.float32, np.float64):
            raise TypeError("dtype must be float32 or float64")

        def dense_cm(a):
            if a is None:
                return None, 1, 0
            if dev:
                import torch
                t = a.to(torch.float32 if dt == np.float32 else torch.float64).t().contiguous()   # row-major transpose == column-major original
                keep.append(t)
                return t.data_ptr(), a.shape[0], a.shape[0]
            arr = np.asfortranarray(np.asarray(a, dtype=dt))
            keep.append(arr)
            return arr.ctypes.data, max(arr.shape[0], 1), arr.shape[0]

        pq, ldq, nq = dense_cm(Sq)
        ps, lds, ns = dense_cm(Ss)
        if dev:
            import torch
            yp, yi, yv = Y[0], Y[1], Y[2]
            nt = int(Y[3])
            # the ABI reads raw pointers: a tensor of another type would be reinterpreted silently -- convert (and keep
            # the converted tensors alive until the constructor has copied them)
            want = torch.float32 if dt == np.float32 else torch.float64
            if not (yp.is_cuda and yi.is_cuda and (yv is None or yv.is_cuda)):
                raise TypeError("device input: Y = (ptr, idx, val, nt) must be CUDA tensors")
            yp = yp.to(torch.int64).contiguous()
            yi = yi.to(torch.int32).contiguous()
            yv = None if yv is None else yv.to(want).contiguous()
            if yp.numel() != ns + 1:
                raise AssertionError("Labels and features have different number of source nodes")
            keep.extend([yp, yi, yv])
            yptr, yidx, yval = yp.data_ptr(), yi.data_ptr(), (None if yv is None else yv.data_ptr())
            mem = L.SS_MEM_DEVICE
        else:
            Y = sp.csr_matrix(Y)
            if Y.shape[0] != ns:
                raise AssertionError("Labels and features have different number of source nodes")
            nt = Y.shape[1]
            parts = _csr_parts(Y, dt)
            keep.append(parts)
            yptr, yidx, yval = _ptr(parts[0]), _ptr(parts[1]), _ptr(parts[2])
            mem = L.SS_MEM_HOST
        h = C.c_void_p()
        if dt == np.float32:
            L.check(lib.ss_graph_create_similarity_f32(nq, ns, nt, pq, ldq, ps, lds, yptr, yidx, yval, 0,
                                                       C.c_float(alpha), 1 if weighted else 0, mem, C.byref(h)))
        else:
            L.check(lib.ss_graph_create_similarity_f64(nq, ns, nt, pq, ldq, ps, lds, yptr, yidx, yval, 0,
                                                       C.c_double(alpha), 1 if weighted else 0, mem, C.byref(h)))
        if dev:
            L.check(lib.ss_synchronize())
        del keep
        return cls(h, dt)

    @classmethod
    def general(cls, A_rows, B, B_cols_T, dtype=np.float64):
        """predict for caller-built A, B (src/core.jl:402-425): A_rows = A[rows, :], B, B_cols_T = B[:, cols]'."""
        import scipy.sparse as sp
        lib = L.lib()
        A_rows, B, Wt = sp.csr_matrix(A_rows), sp.csr_matrix(B), sp.csr_matrix(B_cols_T)
        n = B.shape[0]
        if B.shape[1] != n or A_rows.shape[1] != n or Wt.shape[1] != n:
            raise ValueError("general graph: inconsistent shapes")
        l, b, w = _csr_parts(A_rows, dtype), _csr_parts(B, dtype), _csr_parts(Wt, dtype)
        h = C.c_void_p()
        fn = getattr(lib, f"ss_graph_create_general_{_suffix(dtype)}")
        L.check(fn(n, A_rows.shape[0], Wt.shape[0], _ptr(l[0]), _ptr(l[1]), _ptr(l[2]), _ptr(b[0]), _ptr(b[1]),
                   _ptr(b[2]), _ptr(w[0]), _ptr(w[1]), _ptr(w[2]), 0, L.SS_MEM_HOST, C.byref(h)))
        return cls(h, dtype, general=True)

    # ---------------------------------------------------------------- queries
    def degrees(self):
        """(kf, ks, kt): count degrees of the query-free graph B (src/graphs.jl:9-11, src/core.jl:366)."""
        kf, ks, kt = (np.zeros(n, np.int64) for n in (self.nf, self.ns, self.nt))
        L.check(L.lib().ss_graph_degrees(self._h, _ptr(kf), _ptr(ks), _ptr(kt)))
        return kf, ks, kt

    def _run(self, fn_name, head_args, nrows, clean, out, layout):
        lib = L.lib()
        colmajor = (layout == "col")
        lay = L.SS_LAYOUT_COLMAJOR if colmajor else L.SS_LAYOUT_ROWMAJOR
        shape = (self.nt, nrows) if colmajor else (nrows, self.nt)  # as a C-order buffer
        if out is None:
            out = np.empty(shape, dtype=self.dtype)
        if _is_torch(out):
            if tuple(out.shape) != shape or not out.is_contiguous() or not out.is_cuda:
                raise ValueError(f"out must be a contiguous CUDA tensor of shape {shape}")
            if np.dtype(str(out.dtype).replace("torch.", "")) != self.dtype:
                raise TypeError("out dtype does not match the graph precision")
            ptr, mem = out.data_ptr(), L.SS_MEM_DEVICE
        else:
            if out.shape != shape or out.dtype != self.dtype or not out.flags.c_contiguous:
                raise ValueError(f"out must be a C-contiguous {self.dtype} array of shape {shape}")
            ptr, mem = out.ctypes.data, L.SS_MEM_HOST
        fn = getattr(lib, f"{fn_name}_{self._suf}")
        L.check(fn(self._h, *head_args, 1 if clean else 0, ptr, shape[1], lay, mem))
        if colmajor and not _is_torch(out):
            return out.T  # (nrows, nt) view, Fortran order -- what a Julia caller sees
        return out

    def predict(self, rows: str = "query", row_begin: int = 0, row_end: Optional[int] = None, clean: bool = False,
                out=None, layout: str = "row"):
        """Scores of rows [row_begin,row_end) of the query (or source) nodes against all targets
        (predict, src/core.jl:402-425,446-466; clean! fused, src/core.jl:478-484)."""
        kind = {"query": L.SS_ROWS_QUERY, "source": L.SS_ROWS_SOURCE}[rows]
        limit = self.nq if rows == "query" else self.ns
        row_end = limit if row_end is None else row_end
        return self._run("ss_predict", (kind, row_begin, row_end), row_end - row_begin, clean, out, layout)

    def predict_loo(self, i_begin: int = 0, i_end: Optional[int] = None, clean: bool = False, out=None,
                    layout: str = "row"):
        """Leave-one-out: row i = predict(construct(y, X, [source_i]), y[[source_i], :])."""
        i_end = self.ns if i_end is None else i_end
        return self._run("ss_predict_loo", (i_begin, i_end), i_end - i_begin, clean, out, layout)

    def predict_kfold(self, fold_of_source, nfolds: Optional[int] = None, clean: bool = False):
        """All folds of a k-fold cross-validation in one call: row i = the scores of source i when its fold is
        held out (construct(y, X, members) + predict (+ clean!) for every fold).  Returns (ns, nt) numpy."""
        fold = np.ascontiguousarray(fold_of_source, dtype=np.int32)
        if fold.shape != (self.ns,):
            raise ValueError("fold_of_source must have one entry per source")
        nfolds = int(fold.max()) + 1 if nfolds is None else nfolds
        out = np.empty((self.ns, self.nt), dtype=self.dtype)
        fn = getattr(L.lib(), f"ss_predict_kfold_{self._suf}")
        L.check(fn(self._h, fold.ctypes.data, nfolds, 1 if clean else 0, out.ctypes.data, self.nt,
                   L.SS_LAYOUT_ROWMAJOR, L.SS_MEM_HOST))
        return out

    def close(self):
        if self._h is not None and self._h.value:
            L.load().ss_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSpMat:
    """The sparse operand W of F = W*R on the device (CSR; cut into LDS-tile form on first wide use)."""

    def __init__(self, W, dtype=np.float32):
        import scipy.sparse as sp
        lib = L.lib()
        W = sp.csr_matrix(W)
        self.shape = W.shape
        self.dtype = np.dtype(dtype)
        self._suf = _suffix(dtype)
        p, i, v = _csr_parts(W, dtype)
        self.nnz = int(len(i))
        self._h = C.c_void_p()
        fn = getattr(lib, f"ss_spmat_create_csr_{self._suf}")
        L.check(fn(W.shape[0], W.shape[1], _ptr(p), _ptr(i), _ptr(v), 0, L.SS_MEM_HOST, C.byref(self._h)))

    def cost(self, B: int):
        b, f = C.c_double(), C.c_double()
        L.check(L.load().ss_spmat_cost(self._h, B, C.byref(b), C.byref(f)))
        return b.value, f.value

    def spmm(self, R, out=None, colmajor: bool = False):
        """F = W @ R.  R: (K, B).  colmajor=False: R, F are C-order (K,B)/(M,B) arrays.
        colmajor=True: R is given as its transpose, a C-order (B, K) array, and F comes back as (B, M)."""
        lib = L.lib()
        M, K = self.shape
        torch_in = _is_torch(R)
        if not torch_in:
            R = np.ascontiguousarray(R, dtype=self.dtype)
        elif not R.is_contiguous():
            R = R.contiguous()
        if R.ndim == 1:
            R = R.reshape(-1, 1) if not colmajor else R.reshape(1, -1)
        if colmajor:
            B, k_in = R.shape
        else:
            k_in, B = R.shape
        if k_in != K:
            raise ValueError(f"R has {k_in} rows, W has {K} columns")
        shape = (B, M) if colmajor else (M, B)
        if out is None:
            if torch_in:
                import torch
                out = torch.empty(shape, dtype=R.dtype, device=R.device)
            else:
                out = np.empty(shape, dtype=self.dtype)
        lay = L.SS_LAYOUT_COLMAJOR if colmajor else L.SS_LAYOUT_ROWMAJOR
        if torch_in:
            rp, fp, mem = R.data_ptr(), out.data_ptr(), L.SS_MEM_DEVICE
        else:
            rp, fp, mem = R.ctypes.data, out.ctypes.data, L.SS_MEM_HOST
        fn = getattr(lib, f"ss_spmm_{self._suf}")
        L.check(fn(self._h, rp, B, R.shape[1], lay, fp, shape[1], lay, mem))
        return out

    def close(self):
        if self._h is not None and self._h.value:
            L.load().ss_spmat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def topl(scores, L: int):
    """The L best columns of every row of a score block, in the order ``sortperm(yhat, rev=true)`` gives (score
    descending, ties by ascending column).  `scores`: C-order float32 numpy array or contiguous torch CUDA tensor
    (nrows, ncols).  Returns (idx int32 (nrows, L), val float32 (nrows, L)) of the same kind as the input."""
    lib = L_.lib()
    if _is_torch(scores):
        import torch
        if scores.dtype != torch.float32 or not scores.is_contiguous():
            raise TypeError("scores must be a contiguous float32 tensor")
        nrows, ncols = scores.shape
        idx = torch.empty((nrows, L), dtype=torch.int32, device=scores.device)
        val = torch.empty((nrows, L), dtype=torch.float32, device=scores.device)
        L_.check(lib.ss_topl_f32(scores.data_ptr(), nrows, ncols, ncols, L, idx.data_ptr(), val.data_ptr(), L_.SS_MEM_DEVICE))
        return idx, val
    a = np.ascontiguousarray(scores, dtype=np.float32)
    nrows, ncols = a.shape
    idx = np.empty((nrows, L), np.int32)
    val = np.empty((nrows, L), np.float32)
    L_.check(lib.ss_topl_f32(a.ctypes.data, nrows, ncols, ncols, L, idx.ctypes.data, val.ctypes.data, L_.SS_MEM_HOST))
    return idx, val


def rank_metrics(y, yhat, alpha: float = 20.0) -> dict:
    """AuROC, AuPRC, BEDROC(alpha) and the validity ratio of one score vector, computed on the device
    (src/performance.jl:22-89,558-560).  `y`: labels (non-zero = positive), `yhat`: float32 scores; numpy arrays
    or contiguous torch CUDA tensors (uint8 / float32) of the same length."""
    lib = L_.lib()
    out = (C.c_double * 4)()
    if _is_torch(yhat):
        import torch
        yl = y if (_is_torch(y) and y.dtype == torch.uint8) else (y != 0).to(torch.uint8)
        yl = yl.contiguous().reshape(-1)
        sc = yhat.contiguous().reshape(-1)
        if sc.dtype != torch.float32 or yl.numel() != sc.numel():
            raise TypeError("yhat must be float32 and as long as y")
        L_.check(lib.ss_rank_metrics_f32(yl.data_ptr(), sc.data_ptr(), sc.numel(), float(alpha), out, L_.SS_MEM_DEVICE))
    else:
        sc = np.ascontiguousarray(np.asarray(yhat).ravel(), dtype=np.float32)
        yl = np.ascontiguousarray((np.asarray(y).ravel() != 0).astype(np.uint8))
        if yl.size != sc.size:
            raise AssertionError("The number of scores must be equal to the number of labels")
        L_.check(lib.ss_rank_metrics_f32(yl.ctypes.data, sc.ctypes.data, sc.size, float(alpha), out, L_.SS_MEM_HOST))
    return {"AuROC": out[0], "AuPRC": out[1], "BEDROC": out[2], "validity_ratio": out[3]}


def jaccard_similarity(X, dtype=np.float64):
    """Weighted Jaccard (Ruzicka) similarity between the rows of a feature matrix, on the device: the similarity
    producer of the reference's tutorial (`1 .- pairwise(Jaccard(), X, dims=1)`, docs/src/tutorial/fishers-flowers.jl:66).
    X: (n, d) numpy array or torch CUDA tensor; returns (n, n) of the same kind."""
    lib = L_.lib()
    if _is_torch(X):
        import torch
        if X.dtype not in (torch.float32, torch.float64):
            raise TypeError("X must be float32 or float64")
        suf = "f32" if X.dtype == torch.float32 else "f64"
        n, d = X.shape
        Xc = X.t().contiguous()                    # column-major n x d
        S = torch.empty((n, n), dtype=X.dtype, device=X.device)
        L_.check(getattr(lib, f"ss_similarity_jaccard_{suf}")(Xc.data_ptr(), n, d, n, S.data_ptr(), n, L_.SS_MEM_DEVICE))
        return S                                   # symmetric: row- and column-major coincide
    dt = np.dtype(dtype)
    suf = "f32" if dt == np.float32 else "f64"
    Xf = np.asfortranarray(np.asarray(X, dtype=dt))
    n, d = Xf.shape
    S = np.empty((n, n), dtype=dt, order="F")
    L_.check(getattr(lib, f"ss_similarity_jaccard_{suf}")(Xf.ctypes.data, n, d, max(n, 1), S.ctypes.data, max(n, 1), L_.SS_MEM_HOST))
    return np.ascontiguousarray(S)
