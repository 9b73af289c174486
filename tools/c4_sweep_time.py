"""BASELINE configs[3] as a sweep: per cutoff alpha, the time to create the dense-similarity graph (degree passes over S,
label operands) and to score the first and a following block of 4096 leave-one-out folds (N=50000 by default)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss
from tools.c4_dense import sym_uniform
from tools.c3_loo import rand_csr
n = int(os.environ.get("N", 50000)); nt = 10000; folds = 4096
ss.init(0); ss.use_torch_stream()
gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 4)
S = sym_uniform(n, gen)
yp, yi = rand_csr(n, nt, 0.01, gen)
out = torch.empty((folds, nt), dtype=torch.float32, device="cuda")
for alpha in (0.1, 0.5, 0.9):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = ss.DeviceGraph.from_similarity(None, S, (yp, yi, None, nt), alpha=alpha, weighted=False, dtype=np.float32)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g.predict_loo(0, folds, clean=True, out=out); torch.cuda.synchronize(); t2 = time.perf_counter()
    g.predict_loo(folds, 2 * folds, clean=True, out=out); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(json.dumps({"alpha": alpha, "create_ms": round((t1 - t0) * 1e3, 2), "first_block_ms": round((t2 - t1) * 1e3, 2), "next_block_ms": round((t3 - t2) * 1e3, 2)}))
    g.close()
