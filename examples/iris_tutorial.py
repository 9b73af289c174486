"""The reference's "Fisher's flowers" tutorial (docs/src/tutorial/fishers-flowers.jl) with every numeric step on the
GPU: features -> weighted-Jaccard similarity -> featurize (cutoff) -> leave-one-out SimSpread -> ranked metrics.

    python examples/iris_tutorial.py            # needs an MI355X; data: tests/golden/iris (the reference's files)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simspread_jl_amd as ss  # noqa: E402


def read(name):
    with open(os.path.join(ROOT, "tests", "golden", "iris", name)) as f:
        lines = f.read().splitlines()
    cols = lines[0].split()
    rows = [l.split()[0] for l in lines[1:]]
    return rows, cols, np.array([[float(v) for v in l.split()[1:]] for l in lines[1:]])


def main(alpha: float = 0.9):
    ss.init(0)
    flowers, _, X = read("iris.features")            # 150 x 4 measurements
    _, classes, Y = read("iris.classes")             # 150 x 3 one-hot species
    S = ss.jaccard_similarity(X)                     # 1 .- pairwise(Jaccard(), X, dims=1)
    # featurize(S, alpha, weighted) + construct + predict + clean! for every leave-one-out fold, one resident graph
    out = {}
    for weighted in (True, False):
        g = ss.DeviceGraph.from_dense(None, S, Y, alpha=alpha, weighted=weighted, dtype=np.float64)
        yhat = g.predict_loo(clean=True)             # row i: flower i held out
        valid = yhat[:, 0] != -99
        m = ss.rank_metrics(Y[valid].ravel(), yhat[valid].ravel().astype(np.float32))
        hit = (yhat.argmax(1) == Y.argmax(1))[yhat.max(1) > 0]
        out[weighted] = dict(m, accuracy_of_predicted=float(hit.mean()), predicted=int((yhat.max(1) > 0).sum()))
        print(f"alpha={alpha} weighted={weighted}: AuROC={m['AuROC']:.4f} AuPRC={m['AuPRC']:.4f} "
              f"BEDROC={m['BEDROC']:.4f} validity={m['validity_ratio']:.3f} "
              f"top-1 accuracy on the {out[weighted]['predicted']} flowers with a prediction: {hit.mean():.3f}")
    return out


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 0.9)
