"""BASELINE configs[2] end to end on ONE GPU: 100k x 100k, 1 %, ALL 10^5 leave-one-out folds, scores reduced where
they are produced -- per block of folds the top-L predictions (ss_topl_f32 -> recall@L / precision@L,
src/performance.jl:308-385) and, on sampled blocks, AuROC / AuPRC / BEDROC of the whole block (ss_rank_metrics_f32) --
so the 40 GB score matrix is never gathered or even held.  Reports wall time, folds/s and the stage split.

    python tools/c3_full_loo.py            (BLOCK=2048 L=100 METRIC_EVERY=8 N=100000)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import simspread_jl_amd as ss
from bench import build_c3


def main():
    block = int(os.environ.get("BLOCK", 2048))
    L = int(os.environ.get("L", 100))
    every = int(os.environ.get("METRIC_EVERY", 8))
    ss.init(0)
    ss.use_torch_stream()
    t_build = time.perf_counter()
    g, n = build_c3(ss, torch)
    nfolds = int(os.environ.get("FOLDS", n))
    # labels of the block's rows, dense uint8 (block x n), rebuilt per block from Y's CSR (kept on the device)
    from tools.c3_loo import rand_csr, rand_sym_csr
    gen = torch.Generator(device="cuda"); gen.manual_seed(20250222 + 3)
    rand_sym_csr(n, 0.01, gen)                    # advance the generator exactly as build_c3 did
    yp, yi = rand_csr(n, n, 0.01, gen)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    out = torch.empty((block, n), dtype=torch.float32, device="cuda")
    labels = torch.zeros((block, n), dtype=torch.uint8, device="cuda")
    g.predict_loo(0, block, clean=True, out=out)  # warm: operands are cut at first use
    torch.cuda.synchronize()

    hits = torch.zeros((), dtype=torch.int64, device="cuda")
    positives = torch.zeros((), dtype=torch.int64, device="cuda")
    metrics = []
    ss.timing_hold(True)
    t_topl = t_metric = 0.0
    t0 = time.perf_counter()
    for bi, lo in enumerate(range(0, nfolds, block)):
        hi = min(lo + block, nfolds)
        nb = hi - lo
        g.predict_loo(lo, hi, clean=True, out=out[:nb])
        # labels of these folds
        labels.zero_()
        a, b = int(yp[lo].item()), int(yp[hi].item())
        rows = torch.repeat_interleave(torch.arange(nb, device="cuda"), (yp[lo + 1:hi + 1] - yp[lo:hi]))
        labels[rows, yi[a:b].long()] = 1
        tt = time.perf_counter()
        ti, tv = ss.topl(out[:nb], L)
        hits += labels[:nb].gather(1, ti.long()).sum()
        positives += (b - a)
        torch.cuda.synchronize()
        t_topl += time.perf_counter() - tt
        if every and bi % every == 0:
            tm = time.perf_counter()
            metrics.append(ss.rank_metrics(labels[:nb].reshape(-1), out[:nb].reshape(-1)))
            t_metric += time.perf_counter() - tm
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t = ss.timing_last()
    ss.timing_hold(False)
    res = {"workload": "BASELINE configs[2]: 100k x 100k, 1 %%, full leave-one-out on one MI355X, %d folds in blocks of %d" % (nfolds, block),
           "nnz_X": g.nnz_xs, "nnz_Y": g.nnz_ys, "graph_build_s": round(t_build, 2),
           "wall_s": round(wall, 3), "folds_per_s": round(nfolds / wall, 1), "edges_per_s": nfolds * n / wall,
           "stage1_transfer_s": round(t["transfer_ms"] * 1e-3, 3), "stage2_spmm_s": round(t["spmm_ms"] * 1e-3, 3),
           "epilogue_s": round(t["epilogue_ms"] * 1e-3, 3),
           "topL_reduction_s": round(t_topl, 3), "rank_metrics_s": round(t_metric, 3), "rank_metric_blocks": len(metrics),
           "recall_at_%d" % L: float(hits.item()) / max(1, int(positives.item())),
           "precision_at_%d" % L: float(hits.item()) / (nfolds * L),
           "mean_AuROC_sampled_blocks": float(np.mean([m["AuROC"] for m in metrics])) if metrics else None,
           "bytes_left_the_gpu": nfolds * L * 8, "score_matrix_bytes_never_gathered": nfolds * n * 4}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
