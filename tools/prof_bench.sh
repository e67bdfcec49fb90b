#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (what profiles/rNN/bench_kernel_stats_*.csv hold):
#   gpurun -- 'bash tools/prof_bench.sh'   ->   gpurun_out/prof_bench/{kernel_stats.csv,bench.json}
export TMPDIR=/tmp
OUT=gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cp $(find $OUT/s -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-160
tail -c 600 $OUT/bench.json
