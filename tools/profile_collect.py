"""Copies what tools/profile_all.sh (and tools/pmc_proto.sh <lib-tag>) left under gpurun_out/ into profiles/ and prints the
numbers DESIGN.md section 6 quotes.   python tools/profile_collect.py [tag=r03] [lib-tag=lib3] [scratch dir with
sweep_f64.jsonl / bench.json]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
lib = sys.argv[2] if len(sys.argv) > 2 else "lib3"
scratch = sys.argv[3] if len(sys.argv) > 3 else None
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
for f in ("c2_kernel_stats.csv", "c2_summary.txt", "c3_full_loo.json", "c3_summary.txt", "c4f64_summary.txt",
          "c4f64_timings.jsonl", "c4ring_summary.txt", "c4ring_timings.jsonl", "c5_summary.txt", "sweep_kernel_stats.csv",
          "sweep_summary.txt", "source_sha.txt"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
for f in (f"pmc_{tag}.json", "pmc_latest.json"):
    shutil.copy(os.path.join(ROOT, "gpurun_out", f), os.path.join(dst, f))
libsum = os.path.join(ROOT, "gpurun_out", f"prof_{lib}", "summary.txt")
if os.path.exists(libsum):
    with open(os.path.join(dst, f"{tag}_csell_pmc.txt"), "w") as out:
        out.write("== SQ / LDS / TA counters of the library kernels in tools/sweep.py (SWEEP_B=16,32,64), tools/pmc_proto.sh\n")
        for line in open(libsum):
            out.write(line[:160].rstrip("\n") + "\n")
sha = open(os.path.join(src, "source_sha.txt")).read().split()[-1]
print("source_sha", sha)
for name in ("c2", "c3", "c5"):
    for line in open(os.path.join(dst, f"{tag}_{name}_summary.txt")):
        if ("transfer_kernel" in line or "spmm_sell_kernel" in line) and " calls " in line:
            m = re.search(r"avg\s+([\d.]+) ns", line)
            print(name, line.split("(")[0].strip()[:60], "avg %.3f ms" % (float(m.group(1)) / 1e6))
tail = open(os.path.join(src, "sweep_trace_tail.log")).read()
print("sweep (fp32)", [(int(b), float(ms), float(fr)) for b, ms, fr in
                       re.findall(r'\{\s*"B": (\d+),\s*"ms": ([\d.]+),.*?"frac_hbm": ([\d.]+)', tail, re.S)])
d = json.load(open(os.path.join(dst, f"{tag}_c3_full_loo.json")))
print("c3 full loo", {k: d[k] for k in ("wall_s", "folds_per_s", "stage1_transfer_s", "stage2_spmm_s", "topL_reduction_s", "rank_metrics_s")})
for f in (f"{tag}_c4ring_timings.jsonl", f"{tag}_c4f64_timings.jsonl"):
    for line in open(os.path.join(dst, f)):
        r = json.loads(line)
        print(f, r["weighted"], round(r["transfer_ms"], 2), "ms", round(r["stage1_TFLOPs"], 1), "TF, stage 2", round(r["spmm_ms"], 2))
p = json.load(open(os.path.join(dst, f"pmc_{tag}.json")))
for k, v in p["kernels"].items():
    print("pmc", k[:44], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()
                         if a in ("hbm_bytes_per_launch", "hbm_bytes_unscaled", "l2_request_bytes_per_launch", "l2_hit_rate", "lds_busy_frac", "avg_duration_us")})
if scratch:
    f64 = os.path.join(scratch, "sweep_f64.jsonl")
    if os.path.exists(f64):
        with open(os.path.join(dst, f"{tag}_sweep_f64.jsonl"), "w") as out:
            out.write(json.dumps({"note": f"tools/sweep_f64.py, W 100k x 100k 1 %, fp64, source_sha {sha}; 2-D kernel of round 2 on a box "
                                          "of the same round (SS_CSELL=0): B = 8 / 16 / 32 0.2884 / 0.4849 / 1.1229 ms"}) + "\n")
            out.write(open(f64).read())
        print("sweep (fp64)", [json.loads(l) for l in open(f64)])
    bj = os.path.join(scratch, "bench.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(dst, f"{tag}_bench_final.json"))
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        print("bench", b["ms_per_step"], b["value"], b["roofline"]["avg_launch_ms"], b["roofline_spmm"]["avg_launch_ms"], b["roofline"]["traffic_source"])
        print("bench sweep", [(r["B"], r["ms"], r["frac_hbm"]) for r in b["spmm_narrow_sweep"]["results"]])
        c = b["c3_loo"]
        print("bench c3_loo", {k: c[k] for k in ("folds_per_s", "ms_per_step", "stage1_ms", "stage2_ms", "full_loo_seconds_at_this_rate")})
