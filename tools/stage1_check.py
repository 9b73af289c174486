"""One stage-1 variant (environment given as KEY=VALUE arguments) against the CPU oracle on a small graph; a process per
variant, so that a faulting kernel is named by the last line printed.  usage: python tools/stage1_check.py [chunk=N] K=V ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    os.environ["SS_TRANSFER_CHUNK" if k == "chunk" else k] = v
import numpy as np

import simspread_jl_amd as ss
from oracle import simspread_oracle as O

print("start", sys.argv[1:], flush=True)
ss.init(0)
for weighted in (True, False):
    Xq, Xs, Ys = O.synth_bipartite(257, 3000, 3000, 200, 0.05, 0.03, seed=11, weighted=weighted, dtype=np.float32)
    Xq = Xq.tolil(); Xq[5, :] = 0; Xq = Xq.tocsr()
    Xs = Xs.tolil(); Xs[:, 17] = 0; Xs[:, 18] = 0; Xs = Xs.tocsr()
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    print("  graph built, weighted", weighted, flush=True)
    got = g.predict("query")
    want = O.predict_factored(Xq.astype(np.float64), Xs.astype(np.float64), Ys.astype(np.float64))
    err = np.abs(got - want).max() / np.abs(want).max()
    print("  weighted", weighted, ss.path_last(), "max rel err", err, flush=True)
    assert err < 1e-5
    g.close()
print("ok", sys.argv[1:], flush=True)
