"""Replays a saved fuzz case (FUZZ_SAVE prefix) under subsets of its switches: which switch makes the difference."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import simspread_jl_amd as ss
from oracle import simspread_oracle as O
pre, env = sys.argv[1], json.loads(sys.argv[2])
Xq, Xs, Ys = (sp.load_npz(f"{pre}_{n}.npz") for n in ("xq", "xs", "ys"))
want = O.predict_factored(*(m.astype(np.float64) for m in (Xq, Xs, Ys)), rows="query")
ss.init(0)
def run(e):
    for k in list(os.environ):
        if k.startswith("SS_"): os.environ.pop(k)
    os.environ.update(e)
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=Xq.dtype.type)
    got = np.asarray(g.predict("query"), np.float64)
    d = np.abs(got - want) / np.abs(want).max()
    bad = np.argwhere(d > 1e-5)
    print(json.dumps({"env": e, "err": float(d.max()), "path": ss.path_last(), "bad_cells": int(len(bad)),
                      "bad_rows": sorted(set(int(r) for r in bad[:, 0]))[:12], "bad_cols_n": len(set(int(c) for c in bad[:, 1]))}), flush=True)
    g.close()
run(env)
for k in env:
    run({a: b for a, b in env.items() if a != k})
