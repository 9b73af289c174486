"""Evaluation metrics.  Confusion-matrix metrics: the reference's own known answers
(test/runtests.jl:245-287).  AuROC / AuPRC / BEDROC / validity_ratio: the reference's tests are `skip = true`
(test/runtests.jl:210-223), so the device results are checked against the literal restatement in oracle/
(parity unpinned)."""
import math

import numpy as np
import pytest

import simspread_jl_amd as ss
from oracle import simspread_oracle as oracle


def test_confusion_matrix_metrics_reference_kats():
    tn, fp, fn, tp = 3, 2, 2, 3                                   # test/runtests.jl:246-254
    for f, want in ((ss.f1score, 0.6), (ss.mcc, 0.2), (ss.accuracy, 0.6), (ss.balancedaccuracy, 0.6),
                    (ss.recall, 0.6), (ss.precision, 0.6)):
        assert f(tn, fp, fn, tp) == pytest.approx(want, rel=1e-12)
    y, yhat = [1, 1, 0, 1, 0, 0, 0, 1, 1, 0], [1, 1, 1, 1, 1, 0, 0, 0, 0, 0]   # :256-265, MLBase roc(gt, pred)
    c = ss.roc(y, yhat)
    assert (c.tn, c.fp, c.fn, c.tp) == (3, 2, 2, 3)
    for f, want in ((ss.f1score, 0.6), (ss.mcc, 0.2), (ss.accuracy, 0.6), (ss.balancedaccuracy, 0.6),
                    (ss.recall, 0.6), (ss.precision, 0.6)):
        assert f(c) == pytest.approx(want, rel=1e-12)


def test_mcc_undefined_cases_reference_kats():
    yhat, y = [1, 1, 0, 1, 0, 0, 0, 1, 1, 0], [1, 1, 1, 0, 0, 0, 0, 0, 0, 0]   # test/runtests.jl:279-287
    ones, zeros = np.ones(10, int), np.zeros(10, int)
    for c in (ss.roc(y, ones), ss.roc(y, zeros), ss.roc(ones, yhat), ss.roc(zeros, yhat)):
        assert ss.mcc(c) - ss.mcc(5, 5) < 1e-5
    with pytest.raises(AssertionError):
        ss.accuracy(0, 0, 0, 0)
    assert math.isnan(ss.recall(3, 2, 0, 0)) and math.isnan(ss.precision(3, 0, 2, 0))


def test_oracle_rank_metrics_small_cases():
    # perfect ranking: every positive above every negative
    assert oracle.auroc([1, 1, 0, 0], [0.9, 0.8, 0.2, 0.1]) == pytest.approx(1.0)
    # the trapezoid only spans the thresholds that exist: no (0,0) point (src/performance.jl:53-62)
    assert oracle.auroc([0, 1], [0.9, 0.1]) == pytest.approx(0.0)
    assert oracle.auroc([1, 0, 1, 0], [0.5, 0.5, 0.5, 0.5]) == pytest.approx(0.0)   # one threshold: no area
    assert oracle.validity_ratio([0.0, 1.0, 0.0, 2.0]) == 0.5
    b_good = oracle.bedroc([1, 1, 0, 0, 0, 0, 0, 0], np.arange(8, 0, -1.0))
    b_bad = oracle.bedroc([0, 0, 0, 0, 0, 0, 1, 1], np.arange(8, 0, -1.0))
    assert b_good > 0.9 and b_bad < 0.1
    # roc over thresholds in the mirror agrees with the literal counts
    rng = np.random.default_rng(0)
    y = rng.random(200) < 0.3
    s = np.round(rng.random(200), 1)
    conf = ss.roc(y, s, np.unique(s))
    for t, c in zip(np.unique(s), conf):
        assert c.tp == int((y & (s >= t)).sum()) and c.fp == int((~y & (s >= t)).sum())
    assert ss.maxperformance(y, s, ss.f1score) == max(ss.f1score(c) for c in conf)
    m, sd = ss.meanstdperformance(y, s, ss.accuracy)
    assert m == pytest.approx(ss.meanperformance(y, s, ss.accuracy)) and sd > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,ties", [(1, False), (2, False), (1000, False), (1000, True), (250_000, True)])
def test_device_rank_metrics_against_oracle(n, ties):
    rng = np.random.default_rng(n + ties)
    y = rng.random(n) < 0.2
    if n <= 2:
        y[:] = [True, False][:n]
    s = rng.random(n).astype(np.float32)
    if ties:
        s = np.round(s * 50).astype(np.float32) / 50      # heavy ties, exact zeros
    got = ss.rank_metrics(y, s, alpha=20.0)
    for name, want in (("AuROC", oracle.auroc(y, s)), ("AuPRC", oracle.auprc(y, s)),
                       ("BEDROC", oracle.bedroc(y, s)), ("validity_ratio", oracle.validity_ratio(s))):
        if math.isnan(want):
            assert math.isnan(got[name]), name
        else:
            assert got[name] == pytest.approx(want, rel=1e-9, abs=1e-12), name
    assert ss.AuROC(y, s) == got["AuROC"] and ss.AuPRC(y, s) == got["AuPRC"]       # bitwise repeatable
    assert ss.BEDROC(y, s, rev=False, alpha=5.0) == pytest.approx(oracle.bedroc(y, s, rev=False, alpha=5.0), rel=1e-9)
    assert ss.validity_ratio(s) == got["validity_ratio"]


@pytest.mark.gpu
def test_device_rank_metrics_torch_and_degenerate():
    import torch
    rng = np.random.default_rng(5)
    y = rng.random(5000) < 0.1
    s = rng.random(5000).astype(np.float32)
    got = ss.rank_metrics(torch.from_numpy(y).cuda(), torch.from_numpy(s).cuda())
    assert got["AuROC"] == pytest.approx(oracle.auroc(y, s), rel=1e-9)
    one_class = ss.rank_metrics(np.ones(10), np.arange(10, dtype=np.float32))
    assert math.isnan(one_class["AuROC"])
    with pytest.raises(ss.SimSpreadError):
        ss.rank_metrics(np.zeros(0), np.zeros(0, np.float32))
