"""predict() latency on the C2 graph (10k sources / features / targets, 5 % similarity, 1 % labels) for small and large
query batches: nq = 1 ... 10000, inputs resident, stream-ordered calls timed with HIP events."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss
import bench

ss.init(0)
ss.use_torch_stream()
n = 10_000
Xq, Xs, Ys = bench.synth_c2(n, n, n, n, 0.05, 0.01, seed=20250222 + 2, rank=0)
for nq in [int(x) for x in os.environ.get("NQ", "1,4,16,64,256,1024,4096,10000").split(",")]:
    g = ss.DeviceGraph.from_sparse(Xq[:nq].astype(np.float32), Xs.astype(np.float32), Ys.astype(np.float32), dtype=np.float32)
    out = torch.empty((nq, n), dtype=torch.float32, device="cuda")
    for _ in range(3):
        g.predict("query", out=out)
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ss.timing_hold(True)
    e0.record()
    for _ in range(reps):
        g.predict("query", out=out)
    e1.record()
    torch.cuda.synchronize()
    t = ss.timing_last()
    ss.timing_hold(False)
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"nq": nq, "ms_per_predict": round(ms, 4), "edges_per_s": nq * n / (ms * 1e-3),
                      "stage1_ms": round(t["transfer_ms"] / reps, 4), "stage2_ms": round(t["spmm_ms"] / reps, 4),
                      "kernels": ss.path_last()}))
    g.close()
