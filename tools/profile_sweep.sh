#!/bin/bash
# rocprofv3 passes over the narrow / mid W*R sweep (tools/sweep.py): kernel trace + stats, then FETCH_SIZE, WRITE_SIZE
# (separate passes, MI355X_MICROARCH.md) and the LDS / issue counters.  Summaries land in gpurun_out/prof_<tag>
# (copy summary.txt and the kernel stats into profiles/).
# usage: SWEEP_B=1,4,8,16,32,64 tools/profile_sweep.sh <tag>      (SWEEP_BINARY=1: pattern-only W; SS_COL=0: older kernels)
set -u
TAG=${1:-sweep}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/sweep.py > $OUT/trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/sweep.py > $OUT/pmc_$name.log 2>&1
done
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
