#!/bin/bash
# Partial measurement pass (posterior kernels + configs[4] scoring pass), same recipe as tools/profile_round.sh.
set -e
R=${1:-r02e}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
stats() {
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 "$@" > $OUT/stats_$name.out 2> $OUT/stats_$name.err || true
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  echo "stats $name done"; head -6 $OUT/kernel_stats_$name.csv | cut -c1-200
}
stats posterior tools/prof_workloads.py posterior 10
stats c5 tools/prof_workloads.py c5 5
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$OUT/pmc_posterior/$(echo $C | tr ' ' '_' | cut -c1-40)
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/prof_workloads.py posterior 4 > /dev/null 2> $D.err || true
done
python3 tools/pmc_summary.py $OUT/pmc_posterior gp_posterior_linv > $OUT/pmc_posterior_gp_posterior_linv.txt || true
cat $OUT/pmc_posterior_gp_posterior_linv.txt
grep -h "^posterior:\|^c5:" $OUT/stats_*.out
python3 tools/prof_workloads.py posterior 20 | tail -1
python3 tools/prof_workloads.py c5 10 | tail -1
