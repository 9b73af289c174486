"""ctypes view of oracle/liboracle.so (C restatement, fp64, OpenMP) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_here = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_here, "liboracle.so"))
        _lib.oracle_predict_query.restype = C.c_int
        _lib.oracle_predict_query.argtypes = [C.c_int64] * 4 + [C.c_void_p] * 9 + [C.c_int64, C.c_int64, C.c_void_p, C.c_int]
        _lib.oracle_max_threads.restype = C.c_int
        _lib.oracle_prepare.restype = C.c_void_p
        _lib.oracle_prepare.argtypes = [C.c_int64] * 4 + [C.c_void_p] * 6
        _lib.oracle_predict_rows.restype = C.c_int
        _lib.oracle_predict_rows.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_void_p, C.c_int]
        _lib.oracle_release.restype = None
        _lib.oracle_release.argtypes = [C.c_void_p]
    return _lib


def _parts(m):
    m = sp.csr_matrix(m, dtype=np.float64)
    m.sort_indices()
    return (np.ascontiguousarray(m.indptr, np.int64), np.ascontiguousarray(m.indices, np.int32),
            np.ascontiguousarray(m.data, np.float64))


def predict_query(Xq, Xs, Ys, r0=0, r1=None, threads=0):
    nq, nf = Xq.shape
    ns, nt = Ys.shape
    r1 = nq if r1 is None else r1
    q, s, y = _parts(Xq), _parts(Xs), _parts(Ys)
    out = np.zeros((r1 - r0, nt))
    rc = lib().oracle_predict_query(nq, ns, nf, nt, q[0].ctypes.data, q[1].ctypes.data, q[2].ctypes.data,
                                    s[0].ctypes.data, s[1].ctypes.data, s[2].ctypes.data, y[0].ctypes.data,
                                    y[1].ctypes.data, y[2].ctypes.data, r0, r1, out.ctypes.data, threads)
    assert rc == 0
    return out


class Prepared:
    """The graph-dependent part (transposes, reciprocal degrees) built once; predict() then times the prediction
    only -- what bench.py's cpu_baseline measures beside a GPU path whose operands are already resident."""

    def __init__(self, Xq, Xs, Ys):
        self.nq, self.nf = Xq.shape
        self.ns, self.nt = Ys.shape
        self._q = _parts(Xq)
        s, y = _parts(Xs), _parts(Ys)
        self._h = lib().oracle_prepare(self.nq, self.ns, self.nf, self.nt, s[0].ctypes.data, s[1].ctypes.data,
                                       s[2].ctypes.data, y[0].ctypes.data, y[1].ctypes.data, y[2].ctypes.data)
        assert self._h

    def predict(self, r0=0, r1=None, threads=0, out=None):
        r1 = self.nq if r1 is None else r1
        if out is None:
            out = np.empty((r1 - r0, self.nt))
        q = self._q
        rc = lib().oracle_predict_rows(self._h, q[0].ctypes.data, q[1].ctypes.data, q[2].ctypes.data, r0, r1,
                                       out.ctypes.data, threads)
        assert rc == 0
        return out

    def close(self):
        if self._h:
            lib().oracle_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def max_threads():
    return lib().oracle_max_threads()
