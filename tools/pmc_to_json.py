"""Turn a gpurun_out/prof_<tag> directory (tools/profile.sh) into profiles/pmc_<tag>.json + pmc_latest.json."""
import csv, glob, json, os, sys
from collections import defaultdict
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from simspread_jl_amd import _lib
def avg(counter_dir, counter):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(src, counter_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch, write = avg("pmc_FETCH_SIZE", "FETCH_SIZE"), avg("pmc_WRITE_SIZE", "WRITE_SIZE")
l2req = avg("pmc_TCC_HIT_sum", "TCC_REQ_sum")
l2hit = avg("pmc_TCC_HIT_sum", "TCC_HIT_sum")
ldsact = avg("pmc_SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
grbm = avg("pmc_SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-c3`",
       "source_sha": _lib.source_hash(),
       "units": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies the 128-B requests of wide streaming reads at 64 B, MI355X_MICROARCH.md; for the 2-4 byte-per-lane loads of transfer_kernel the factor is uncalibrated, so hbm_bytes_unscaled = (FETCH_SIZE + WRITE_SIZE)*1024 is given too; both count Infinity-Cache hits as memory traffic)",
       "kernels": {}}
for k in fetch:
    if not any(s in k for s in ("transfer", "spmm", "reduce_kernel", "unpermute")):
        continue
    out["kernels"][k] = {"FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k),
                         "hbm_bytes_per_launch": (2 * fetch[k] + write.get(k, 0)) * 1024,
                         "hbm_bytes_unscaled": (fetch[k] + write.get(k, 0)) * 1024,
                         "l2_request_bytes_per_launch": l2req.get(k, 0) * 64,
                         "l2_hit_rate": (l2hit.get(k, 0) / l2req[k]) if l2req.get(k) else None,
                         "lds_busy_frac": ((ldsact.get(k, 0) / 256) / (grbm[k] / 8)) if grbm.get(k) else None,
                         "avg_duration_us": (sum(dur[k]) / len(dur[k]) / 1e3) if k in dur else None}
# profiles/ is what gets committed; on a GPU box only gpurun_out/ travels back, so a copy goes there too
for d in ("profiles", "gpurun_out"):
    if os.path.isdir(os.path.join(root, d)):
        for name in (f"pmc_{tag}.json", "pmc_latest.json"):
            with open(os.path.join(root, d, name), "w") as f:
                json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
