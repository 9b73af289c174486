"""Wall time of DeviceGraph.from_sparse (host CSR -> resident graph: copies, validation, transposes, degrees), of the first
predict (operands are cut at first use) and of a steady-state predict on the C2 inputs."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss
import bench
ss.init(0); ss.use_torch_stream()
n = 10_000
Xq, Xs, Ys = bench.synth_c2(n, n, n, n, 0.05, 0.01, seed=20250222 + 2, rank=0)
Xq, Xs, Ys = [m.astype(np.float32) for m in (Xq, Xs, Ys)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = torch.empty((n, n), dtype=torch.float32, device="cuda")
    g.predict("query", out=out); torch.cuda.synchronize(); t2 = time.perf_counter()
    g.predict("query", out=out); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(json.dumps({"create_ms": (t1-t0)*1e3, "first_predict_ms": (t2-t1)*1e3, "second_predict_ms": (t3-t2)*1e3}))
    g.close()
