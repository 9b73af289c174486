"""Evaluation metrics of the reference (src/performance.jl) behind the reference's names.

Confusion-matrix metrics are host arithmetic on four integers; the threshold-free ones (AuROC, AuPRC, BEDROC,
validity_ratio) run on the device through ``ss_rank_metrics_f32`` so that a score block can be judged where it
was produced.  ``roc`` stands in for MLBase's ``roc``/``ROCNums`` (MLBase v0.9: a sample is predicted positive
when ``score >= threshold``); the reference's own tests for the threshold-free metrics are ``skip = true``
(test/runtests.jl:210-223), so for those the reference source is the only definition (parity unpinned).
"""
from __future__ import annotations

import math
from collections import namedtuple
from typing import Callable, List, Sequence

import numpy as np

ROCNums = namedtuple("ROCNums", "p n tp tn fp fn")


def roc(y, yhat, thresholds=None):
    """roc(gt, pred) for 0/1 predictions, or roc(gt, scores, thresholds) -> list of ROCNums (one per threshold,
    positive when score >= threshold)."""
    y = np.asarray(y).ravel() != 0
    yhat = np.asarray(yhat).ravel()
    if len(y) != len(yhat):
        raise AssertionError("The number of scores must be equal to the number of labels")
    p, n = int(y.sum()), int((~y).sum())

    def nums(pred):
        tp = int((pred & y).sum()); fp = int((pred & ~y).sum())
        return ROCNums(p, n, tp, n - fp, fp, p - tp)

    if thresholds is None:
        return nums(yhat != 0)
    if np.isscalar(thresholds):
        return nums(yhat >= thresholds)
    # all thresholds at once: sort the scores descending, count by searchsorted
    order = np.argsort(-yhat, kind="stable")
    ys, ss = y[order], yhat[order]
    ctp = np.concatenate([[0], np.cumsum(ys)])
    out = []
    neg = -ss  # ascending
    for t in np.asarray(thresholds).ravel():
        k = int(np.searchsorted(neg, -t, side="right"))  # scores >= t
        tp = int(ctp[k]); fp = k - tp
        out.append(ROCNums(p, n, tp, n - fp, fp, p - tp))
    return out


def _four(args):
    if len(args) == 1:
        c = args[0]
        return int(c.tn), int(c.fp), int(c.fn), int(c.tp)
    tn, fp, fn, tp = (int(a) for a in args)
    return tn, fp, fn, tp


def _nonempty(tn, fp, fn, tp):
    if not tn + fp + fn + tp > 0:
        raise AssertionError("Confusion matrix sums zero!")


def f1score(*args) -> float:
    """f1score(tn, fp, fn, tp) | f1score(confusion)   (src/performance.jl:102-114)"""
    tn, fp, fn, tp = _four(args)
    _nonempty(tn, fp, fn, tp)
    d = tp + 0.5 * (fp + fn)
    return math.nan if d == 0 else tp / d


_FLOATMIN = 2.2250738585072014e-308


def mcc(*args, eps: float = _FLOATMIN) -> float:
    """mcc(a, b[, eps]) limit form | mcc(tn, fp, fn, tp) | mcc(confusion)   (src/performance.jl:148-190)"""
    if len(args) in (2, 3) and not hasattr(args[0], "tn"):
        a, b = float(args[0]), float(args[1])
        e = float(args[2]) if len(args) == 3 else eps
        return (a * e - b * e) / math.sqrt((a + b) * (a + e) * (b + e) * (e + e))
    tn, fp, fn, tp = _four(args)
    _nonempty(tn, fp, fn, tp)
    p_pred, n_pred, p_act, n_act = tp + fp, fn + tn, tp + fn, fp + tn
    if p_pred == 0:
        return mcc(tn, fn)
    if n_pred == 0:
        return mcc(tp, fp)
    if p_act == 0:
        return mcc(tn, fp)
    if n_act == 0:
        return mcc(tp, fn)
    return ((tp * tn) - (fp * fn)) / math.sqrt(p_pred * n_pred * p_act * n_act)


def accuracy(*args) -> float:
    tn, fp, fn, tp = _four(args)
    _nonempty(tn, fp, fn, tp)
    return (tp + tn) / ((tp + tn) + (fp + fn))


def balancedaccuracy(*args) -> float:
    tn, fp, fn, tp = _four(args)
    _nonempty(tn, fp, fn, tp)
    tpr = tp / (tp + fn) if tp + fn else math.nan
    tnr = tn / (tn + fp) if tn + fp else math.nan
    return (tpr + tnr) / 2


def recall(*args) -> float:
    tn, fp, fn, tp = _four(args)
    _nonempty(tn, fp, fn, tp)
    return math.nan if tp + fn == 0 else tp / (tp + fn)


def precision(*args) -> float:
    tn, fp, fn, tp = _four(args)
    _nonempty(tn, fp, fn, tp)
    return math.nan if tp + fp == 0 else tp / (tp + fp)


# ------------------------------------------------------------------ threshold-free metrics (device)
def _rank(y, yhat, alpha=20.0):
    from .engine import rank_metrics
    if np.size(y) != np.size(yhat):
        raise AssertionError("The number of scores must be equal to the number of labels")
    return rank_metrics(y, yhat, alpha)


def AuROC(y, yhat) -> float:
    """Area under the ROC curve, trapezoidal rule over the unique-score thresholds (src/performance.jl:49-63)."""
    return _rank(y, yhat)["AuROC"]


def AuPRC(y, yhat) -> float:
    """Area under the precision-recall curve (src/performance.jl:74-89)."""
    return _rank(y, yhat)["AuPRC"]


def BEDROC(y, yhat, rev: bool = True, alpha: float = 20.0) -> float:
    """BEDROC(y, yhat; rev=true, alpha=20.0)   (src/performance.jl:22-38).  rev=False ranks ascending, which is the
    stable descending order of the negated scores."""
    s = yhat if rev else -(yhat if hasattr(yhat, "device") else np.asarray(yhat, dtype=np.float32))
    return _rank(y, s, alpha)["BEDROC"]


def validity_ratio(yhat) -> float:
    """Share of non-zero predictions (src/performance.jl:558-560)."""
    n = int(np.prod(yhat.shape)) if hasattr(yhat, "shape") else len(yhat)
    return _rank(np.zeros(n, np.uint8) if not hasattr(yhat, "device") else (yhat != 0), yhat)["validity_ratio"]


# ------------------------------------------------------------------ metric over all thresholds (host)
def _confusions(y, yhat) -> List[ROCNums]:
    yhat = np.asarray(yhat).ravel()
    return roc(y, yhat, np.unique(yhat))


def maxperformance(y, yhat, metric: Callable) -> float:
    """src/performance.jl:420-438"""
    conf = y if yhat is None else _confusions(y, yhat)
    return max(metric(c) for c in conf)


def meanperformance(y, yhat, metric: Callable) -> float:
    """src/performance.jl:447-478"""
    conf = y if yhat is None else _confusions(y, yhat)
    return float(np.mean([metric(c) for c in conf]))


def meanstdperformance(y, yhat, metric: Callable):
    """mean and (n-1) standard deviation, StatsBase.mean_and_std (src/performance.jl:487-520)"""
    conf = y if yhat is None else _confusions(y, yhat)
    v = np.array([metric(c) for c in conf], dtype=np.float64)
    return float(v.mean()), float(v.std(ddof=1)) if len(v) > 1 else math.nan
