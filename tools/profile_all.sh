#!/bin/bash
# Every profile the round's DESIGN / bench numbers quote, from ONE build of the kernels (the source hash goes into each
# summary): C2 step (trace + counters + FETCH/WRITE), memory-path counters of the C2 kernels, C3 block, C3 full LOO,
# C4 bf16 ring at 50k, C4 fp64 at 20k, C5 block, narrow / mid sweep.  Each rocprofv3 pass is bounded by `timeout` and
# logs its progress, the program itself follows `--`.
# usage: tools/profile_all.sh <tag>        -> gpurun_out/prof_<tag>_*/ ; copy with tools/profile_collect.py
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# raw rocprofv3 output (kernel traces: tens of MB) stays in /tmp on the box; summaries, logs and the small JSON files go to
# gpurun_out/prof_<tag>/ (gpurun copies back at most 64 MiB)
P=/tmp/prof_${TAG}
S=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}
rm -rf $P ${P}_*
mkdir -p $P $S
LOG=$S/progress.txt
: > $LOG
SHA=$(python3 -c "import simspread_jl_amd as s; print(s._lib.source_hash())")
echo "source_sha $SHA" | tee -a $LOG
pass() {   # pass <dir> <counters or ""> -- program args...
  local d=$1; local grp=$2; shift 2
  echo "$(date +%T) $d [$grp]" >> $LOG
  if [ -z "$grp" ]; then
    timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- "$@" > $d.log 2>&1
  else
    timeout -k 5 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- "$@" > $d.log 2>&1
  fi
  local rc=$?
  echo "   rc $rc" >> $LOG
  if [ $rc -ne 0 ]; then tail -5 $d.log >> $LOG; fi
}
BENCH="python3 bench.py --no-sweep --no-cpu-baseline --no-c3 --no-c5"
# ---- C2
mkdir -p ${P}_c2
pass ${P}_c2/trace "" $BENCH --steps 100 --warmup 5
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  pass ${P}_c2/pmc_$name "$grp" $BENCH --steps 5 --warmup 2
done
python3 tools/prof_summary.py ${P}_c2 > $S/c2_summary.txt 2>&1; cp ${P}_c2/trace/*/*kernel_stats.csv $S/c2_kernel_stats.csv 2>/dev/null
python3 tools/pmc_to_json.py ${P}_c2 $TAG > $S/pmc_to_json.log 2>&1
# ---- C3 block
mkdir -p ${P}_c3
export CHECK=0
pass ${P}_c3/trace "" python3 tools/c3_loo.py
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  pass ${P}_c3/pmc_$name "$grp" python3 tools/c3_loo.py
done
python3 tools/prof_summary.py ${P}_c3 > $S/c3_summary.txt 2>&1
echo "$(date +%T) c3 full loo" >> $LOG
timeout -k 5 300 python3 tools/c3_full_loo.py > $S/c3_full_loo.json 2> $S/c3_full_loo.err
# ---- C5 block
mkdir -p ${P}_c5
pass ${P}_c5/trace "" python3 tools/c5_powerlaw.py
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  pass ${P}_c5/pmc_$name "$grp" python3 tools/c5_powerlaw.py
done
python3 tools/prof_summary.py ${P}_c5 > $S/c5_summary.txt 2>&1
# ---- C4: bf16 ring kernel at 50k, fp64 kernel at 20k (one alpha, both weightings)
for cfg in "50000 f32 c4ring" "20000 f64 c4f64"; do
  set -- $cfg
  export N=$1 DTYPE=$2 ALPHAS=0.1
  mkdir -p ${P}_$3
  pass ${P}_$3/trace "" python3 tools/c4_dense.py
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | cut -d' ' -f1)
    pass ${P}_$3/pmc_$name "$grp" python3 tools/c4_dense.py
  done
  timeout -k 5 300 python3 tools/c4_dense.py > $S/$3_timings.jsonl 2> $S/$3_timings.err
  python3 tools/prof_summary_c4.py ${P}_$3 > $S/$3_summary.txt 2>&1
done
unset N DTYPE ALPHAS
# ---- narrow / mid sweep
mkdir -p ${P}_sweep
pass ${P}_sweep/trace "" python3 tools/sweep.py
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  pass ${P}_sweep/pmc_$name "$grp" python3 tools/sweep.py
done
python3 tools/prof_summary.py ${P}_sweep > $S/sweep_summary.txt 2>&1; cp ${P}_sweep/trace/*/*kernel_stats.csv $S/sweep_kernel_stats.csv 2>/dev/null; tail -c 6000 ${P}_sweep/trace.log > $S/sweep_trace_tail.log
echo "source_sha $SHA" > $S/source_sha.txt
echo "$(date +%T) done" >> $LOG
