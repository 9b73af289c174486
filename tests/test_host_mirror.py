"""Host logic of the mirror that needs no GPU: construct's node order, name filter and assertion messages
(test/runtests.jl:83-111), NamedMatrix semantics, the dense view of a Network against the oracle's literal blocks."""
import numpy as np
import pytest

import simspread_jl_amd as ss
from oracle import simspread_oracle as O


def _kat_inputs(kats):
    c = kats["construct"]
    return c, ss.NamedMatrix(c["X"], c["X_rows"], c["X_cols"]), ss.NamedMatrix(c["y"], c["y_rows"], c["y_cols"])


def test_construct_node_order_and_messages(kats):
    c, X, y = _kat_inputs(kats)
    A, B = ss.construct(y, X, c["queries"])
    assert A.names(1) == A.names(2) == c["node_order"]
    assert B.names(1) == B.names(2) == c["node_order"]
    with pytest.raises(AssertionError, match=c["same_names_message"]):
        ss.construct(y, ss.NamedMatrix(c["X"], c["X_rows"], c["X_rows"]), c["queries"])
    with pytest.raises(AssertionError, match=c["row_mismatch_message"]):
        ss.construct(y, ss.NamedMatrix(c["row_mismatch_X"], c["row_mismatch_rows"], c["X_cols"]), c["queries"])


def test_construct_dense_views_equal_the_literal_blocks(kats):
    c, X, y = _kat_inputs(kats)
    A, B = ss.construct(y, X, c["queries"])
    Ao, Bo = O.construct_queries(O.Named(c["y"], c["y_rows"], c["y_cols"]), O.Named(c["X"], c["X_rows"], c["X_cols"]),
                                 c["queries"])
    np.testing.assert_array_equal(A.array, Ao.array)
    np.testing.assert_array_equal(B.array, Bo.array)
    # split form and 3-layer form
    rng = np.random.default_rng(0)
    tr, te = [f"a{i}" for i in range(6)], [f"b{i}" for i in range(3)]
    tn = ["t0", "t1"]
    Xtr = ss.NamedMatrix(rng.random((6, 6)), tr, ["f" + n for n in tr]); Xte = ss.NamedMatrix(rng.random((3, 6)), te, ["f" + n for n in tr])
    ytr = ss.NamedMatrix(rng.integers(0, 2, (6, 2)), tr, tn); yte = ss.NamedMatrix(rng.integers(0, 2, (3, 2)), te, tn)
    A2, B2 = ss.construct(ytr, yte, Xtr, Xte)
    Ao2, Bo2 = O.construct_split(O.Named(ytr.array, tr, tn), O.Named(yte.array, te, tn),
                                 O.Named(Xtr.array, tr, Xtr.cols), O.Named(Xte.array, te, Xte.cols))
    assert A2.names(1) == Ao2.rows
    np.testing.assert_array_equal(A2.array, Ao2.array)
    np.testing.assert_array_equal(B2.array, Bo2.array)
    A3 = ss.construct(ytr, Xtr)
    np.testing.assert_array_equal(A3.array, O.construct_single(O.Named(ytr.array, tr, tn), O.Named(Xtr.array, tr, Xtr.cols)).array)
    with pytest.raises(AssertionError, match="Number of targets between test and training sets doesn't match"):
        ss.construct(ytr, ss.NamedMatrix(np.zeros((3, 3)), te, ["t0", "t1", "t2"]), Xtr, Xte)
    with pytest.raises(AssertionError, match="Number of features between test and training sets doesn't match"):
        ss.construct(ytr, yte, Xtr, ss.NamedMatrix(np.zeros((3, 5)), te, [f"g{i}" for i in range(5)]))


def test_name_filter_and_name_check_quirks():
    names = ["foo", "bar", "oo"]
    X = ss.NamedMatrix(np.ones((3, 3)), names, ["f" + n for n in names])
    y = ss.NamedMatrix(np.eye(3), names, ["t0", "t1", "t2"])
    A, _ = ss.construct(y, X, ["oo"])      # lstrip('f') strips every leading f: "ffoo" and "foo" both vanish
    assert A.names(1) == ["oo", "foo", "bar", "fbar", "t0", "t1", "t2"]
    with pytest.raises(ValueError, match="DimensionMismatch"):   # sorted-vector compare needs Nf == Ns
        ss.construct(ss.NamedMatrix(np.ones((3, 1)), names, ["t"]), ss.NamedMatrix(np.ones((3, 2)), names, ["fx", "fy"]))


def test_named_matrix_semantics():
    M = ss.NamedMatrix([[1, 2], [3, 4]])
    assert M.names(1) == ["1", "2"] and M.names(2) == ["1", "2"]       # NamedArrays default names
    assert M == ss.NamedMatrix([[1.0, 2.0], [3.0, 4.0]], ["a", "b"], ["c", "d"])   # == compares values only
    assert M.sub(["2"], ["2", "1"]).array.tolist() == [[4.0, 3.0]]
    with pytest.raises(ValueError):
        ss.NamedMatrix(np.zeros((2, 2)), ["a", "a"], ["c", "d"])
    with pytest.raises(KeyError):
        M.sub(["9"], ["1"])
