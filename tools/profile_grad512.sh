export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03e; mkdir -p $OUT
python3 tools/dev_grad512_time.py 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_g512 -- python3 tools/dev_grad512_time.py > $OUT/g512.out 2> $OUT/g512.err || true
find $OUT/stats_g512 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_g512.csv \;
head -8 $OUT/kernel_stats_g512.csv | cut -c1-160
