"""Narrow-B W*R sweep on its own (same code path as bench.py's spmm_narrow_sweep) with wall-clock cross-check."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss
import bench
ss.init(0)
ss.use_torch_stream()
t0 = time.perf_counter()
out = bench.spmm_sweep(ss, torch, steps=int(os.environ.get("SWEEP_STEPS", "5")))
print(json.dumps(out, indent=0))
print("sweep wall", time.perf_counter() - t0)
