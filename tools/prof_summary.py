"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into one text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print("%-90s calls %6s  total %12s ns  avg %12s ns  %6s%%" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))

print("\n== per-kernel averages from the kernel trace (dispatches after warm-up included)")
for f in find("trace/**/*kernel_trace.csv"):
    agg = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:8]:
        d = [x[0] for x in v]
        print("%-80s n=%4d avg %10.1f us min %10.1f us  vgpr %s lds %s grid %s wg %s" % (k[:80], len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, v[0][1], v[0][2], v[0][3], v[0][4]))

print("\n== PMC (average per dispatch of each kernel)")
for d in find("pmc_*/"):
    for f in find(os.path.join(os.path.basename(os.path.normpath(d)), "**/*counter_collection.csv")):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if not any(s in k for s in ("transfer", "spmm", "reduce_kernel", "unpermute", "csell")):
                continue
            print(k[:70])
            for cn, vals in sorted(cs.items()):
                print("    %-28s %16.1f  (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
