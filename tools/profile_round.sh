#!/bin/bash
# Measurement pass for one round on the GPU box: bench line, rocprofv3 kernel stats, PMC passes.
# usage (via gpurun): bash tools/profile_round.sh r01
set -e
R=${1:-r01}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2> $D.err
  echo "pmc $C done"
done
python3 tools/pmc_summary.py $OUT gp_fit_fused > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -5 $OUT/kernel_stats.csv
