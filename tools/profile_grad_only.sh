set -e
R=r02d
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_grad -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --step fit+grad > $OUT/stats_grad.out 2> $OUT/stats_grad.err || true
find $OUT/stats_grad -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_grad.csv \;
head -4 $OUT/kernel_stats_grad.csv | cut -c1-160
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$OUT/pmc_grad/$(echo $C | tr ' ' '_' | cut -c1-40)
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --step fit+grad > /dev/null 2> $D.err || true
done
python3 tools/pmc_summary.py $OUT/pmc_grad gp_mll_grad_fused > $OUT/pmc_grad_gp_mll_grad_fused.txt || true
cat $OUT/pmc_grad_gp_mll_grad_fused.txt
