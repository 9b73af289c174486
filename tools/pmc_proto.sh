#!/bin/bash
# SQ / LDS counters of the prototype kernel (tools/csell_proto.py): tools/pmc_proto.sh <tag> <B...>
set -u
TAG=${1:-proto}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  echo "pass $i: $grp" >> $OUT/progress.txt
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $RAW/pmc_g$i -- python3 ${PMC_SCRIPT:-tools/csell_proto.py} "$@" > $OUT/pmc_g$i.log 2>&1
  echo "   rc $?" >> $OUT/progress.txt
done
python3 tools/prof_summary.py $RAW > $OUT/summary.txt 2>&1
grep -A12 "csell\|spmm_colgroup_kernel" $OUT/summary.txt | cut -c1-120
