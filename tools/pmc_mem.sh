#!/bin/bash
# Memory-path counters (TA / TCP / TD / address translation) of the C2 step, one rocprofv3 pass per group.
# usage: tools/pmc_mem.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-mem}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARGS="--steps 5 --warmup 2 --no-sweep --no-cpu-baseline --no-c3 --no-c5 $*"
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $grp" >> $OUT/progress.txt
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_g$i -- python3 bench.py $ARGS > $OUT/pmc_g$i.log 2>&1
  echo "   rc $?" >> $OUT/progress.txt
done
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
