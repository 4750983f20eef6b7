#!/bin/bash
# PMC passes for the kernels of the blocked fit (tools/dev_fit512_time.py 32,512,6), same recipe as tools/profile_round.sh.
set -e
R=${1:-r02h}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$OUT/pmc_blocked/$(echo $C | tr ' ' '_' | cut -c1-40)
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 tools/dev_fit512_time.py 32,512,6 > /dev/null 2> $D.err || true
done
for K in gp_blocked_solve gp_blocked_syrk scaml_blocked_finish gp_fit_blocked_kernel; do
  python3 tools/pmc_summary.py $OUT/pmc_blocked $K > $OUT/pmc_blocked_$K.txt || true
  echo "== $K"; cat $OUT/pmc_blocked_$K.txt
done
