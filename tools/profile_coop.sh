#!/bin/bash
# rocprofv3 kernel stats + PMC passes of the several-CUs-per-task fit (csrc/gp_fit_coop.hip) at the configs[4] source stack.
# usage (via gpurun): bash tools/profile_coop.sh r03d      (outputs under gpurun_out/r03d; copy the summaries to profiles/)
set -e
R=${1:-r03d}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_coop -- python3 tools/dev_coop_time.py 32,512,6 > $OUT/stats_coop.out 2> $OUT/stats_coop.err || true
find $OUT/stats_coop -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_coop.csv \;
head -8 $OUT/kernel_stats_coop.csv | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$OUT/pmc_coop/$(echo $C | tr ' ' '_' | cut -c1-40)
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/dev_coop_time.py 32,512,6 > /dev/null 2> $D.err || true
done
python3 tools/pmc_summary.py $OUT/pmc_coop gp_fit_coop > $OUT/pmc_coop_gp_fit_coop.txt || true
cat $OUT/pmc_coop_gp_fit_coop.txt
