set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c3
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  CHECK=0 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 tools/c3_loo.py > $OUT/pmc_$name.log 2>&1
done
python3 tools/prof_summary.py $OUT 2>&1 | grep -A8 "spmm_sell\|transfer_kernel" | head -60
