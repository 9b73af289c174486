"""A/B timing of stage 1 (transfer kernel) on the C2 workload under the library's tuning switches.
Each variant builds a fresh graph handle (the chunked operand is cut at first use, under the variant's
environment), scores the same queries, reports the per-launch stage times and compares the scores with the
first variant's (max relative difference and bitwise equality).

    python tools/stage1_ab.py "SS_CHUNK_SCHED=0" "SS_CHUNK_SCHED=1" "SS_CHUNK_SCHED=1 SS_TRANSFER_DUAL=1"
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import simspread_jl_amd as ss
from bench import synth_c2

KEYS = ("SS_CHUNK_SCHED", "SS_TRANSFER_DUAL", "SS_TRANSFER_CHUNK", "SS_TRANSFER_U", "SS_TRANSFER_QB", "SS_TRANSFER_V",
        "SS_TRANSFER_ALIGN", "SS_TRANSFER_NW", "SS_TRANSFER_X", "SS_TRANSFER_FIX", "SS_TRANSFER_FIX1", "SS_TRANSFER_PIPE", "SS_TRANSFER_RING", "SS_TRANSFER_FLAT", "SS_TRANSFER_LD", "SS_TRANSFER_QFLAT", "SS_TRANSFER_WIDE", "SS_TRANSFER_DBG", "SS_TRANSFER_WIDE2")


def main():
    variants = sys.argv[1:] or ["SS_CHUNK_SCHED=0", "SS_CHUNK_SCHED=1"]
    n = int(os.environ.get("N", 10_000))
    steps = int(os.environ.get("STEPS", 20))
    weighted = os.environ.get("UNWEIGHTED", "0") != "1"
    ss.init(0)
    ss.use_torch_stream()
    Xq, Xs, Ys = synth_c2(n, n, n, n, 0.05, 0.01, seed=20250222 + 2, rank=0, weighted=weighted)
    ref = None
    rounds = int(os.environ.get("ROUNDS", 1))   # ROUNDS > 1: the variants are run round-robin (same process), per-variant min / median
    for v in variants * rounds:
        for k in KEYS:
            os.environ.pop(k, None)
        for kv in v.split():
            k, val = kv.split("=")
            os.environ[k] = val
        g = ss.DeviceGraph.from_sparse(Xq, Xs, Ys, dtype=np.float32)
        out = torch.empty((n, n), dtype=torch.float32, device="cuda")
        for _ in range(3):
            g.predict("query", out=out)
        torch.cuda.synchronize()
        ss.timing_hold(True)
        for _ in range(steps):
            g.predict("query", out=out)
        torch.cuda.synchronize()
        t = ss.timing_last()
        ss.timing_hold(False)
        res = {"variant": v, "path": ",".join(ss.path_last()), "transfer_ms": round(t["transfer_ms"] / t["transfer_launches"], 4),
               "spmm_ms": round(t["spmm_ms"] / t["spmm_launches"], 4)}
        if ref is None:
            ref = out.clone()
        else:
            res["bitwise_equal_to_first"] = bool(torch.equal(out, ref))
            res["max_rel_diff_to_first"] = float(((out - ref).abs().max() / ref.abs().max()).item())
        print(json.dumps(res), flush=True)
        g.close()


if __name__ == "__main__":
    main()
