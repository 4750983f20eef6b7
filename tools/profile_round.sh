#!/bin/bash
# Measurement pass for one round on the GPU box: bench line, rocprofv3 kernel stats for every hot path (fit, fit + gradient,
# posterior, configs[4] scoring pass), PMC passes per kernel, FETCH_SIZE / WRITE_SIZE calibration.
# usage (via gpurun): bash tools/profile_round.sh r02        (outputs under gpurun_out/r02; copy the summaries to profiles/)
set -e
R=${1:-r03}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; cat $OUT/bench.json
stats() {   # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 "$@" > $OUT/stats_$name.out 2> $OUT/stats_$name.err || true
  find $OUT/stats_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  echo "stats $name done"; head -6 $OUT/kernel_stats_$name.csv | cut -c1-200
}
stats fit bench.py --steps 100 --warmup 10 --no-cpu-baseline
stats grad bench.py --steps 50 --warmup 10 --no-cpu-baseline --step fit+grad
stats posterior tools/prof_workloads.py posterior 10
stats c5 tools/prof_workloads.py c5 5
stats blocked tools/dev_fit512_time.py 32,512,6
python3 tools/dev_fit512_time.py > $OUT/blocked_fit_timings.txt 2>&1 || true
cat $OUT/blocked_fit_timings.txt
[ -x tools/dp_pipe_probe ] && tools/dp_pipe_probe > $OUT/probe_dp_pipe.txt 2>&1 || true
pmc() {   # name, kernel substring, program args...
  local name=$1 needle=$2; shift; shift
  for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    local D=$OUT/pmc_${name}/$(echo $C | tr ' ' '_' | cut -c1-40)
    mkdir -p $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 "$@" > /dev/null 2> $D.err || true
  done
  python3 tools/pmc_summary.py $OUT/pmc_${name} $needle > $OUT/pmc_${name}_${needle}.txt || true
  echo "pmc $name / $needle done"; cat $OUT/pmc_${name}_${needle}.txt
}
pmc fit gp_fit_fused bench.py --steps 20 --warmup 2 --no-cpu-baseline
python3 tools/pmc_summary.py $OUT/pmc_fit gp_fit_fused --json $OUT/pmc_traffic.json > /dev/null || true
pmc grad gp_mll_grad_fused bench.py --steps 10 --warmup 2 --no-cpu-baseline --step fit+grad
pmc posterior gp_posterior_linv tools/prof_workloads.py posterior 4
pmc c5 scaml_target_fit tools/prof_workloads.py c5 3
python3 tools/pmc_summary.py $OUT/pmc_posterior gp_linv_kernel > $OUT/pmc_posterior_gp_linv_kernel.txt || true
python3 tools/pmc_summary.py $OUT/pmc_posterior gp_posterior_kernel > $OUT/pmc_posterior_gp_posterior_kernel.txt || true
# calibration of the two traffic counters for 8-byte and 16-byte per-lane streams (tools/fetch_calib.hip)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -- tools/fetch_calib > $OUT/calib_$C.out 2> $OUT/calib_$C.err || true
done
for K in read8 read16 write8 write16 write8_rowseg; do
  echo "== $K"; python3 tools/pmc_summary.py $OUT/calib_FETCH_SIZE "$K(" ; python3 tools/pmc_summary.py $OUT/calib_WRITE_SIZE "$K("
done > $OUT/calib_summary.txt 2>&1 || true
cat $OUT/calib_summary.txt
