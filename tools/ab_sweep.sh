#!/bin/bash
# Config E (1 M points, 0.05 m cells): cells per workgroup round of the byte-mask sweep (AVL_SWEEP_VEC = 4 / 2 / 1 -> 16384 / 8192 /
# 4096), experiments build; tools/bench_mapping.py prints GPU microseconds per frame for configs C and E (f64 SoA and f32 AoS clouds).
. tools/use_experiments_lib.sh
OUT=${1:-gpurun_out/r4}
mkdir -p $OUT
for v in ${VECS:-4 2 1 4 2 1}; do
  echo "=== AVL_SWEEP_VEC=$v" | tee -a $OUT/ab_sweep.log
  AVL_SWEEP_VEC=$v timeout -k 10 200 python tools/bench_mapping.py 2>&1 | grep "us/frame" | tee -a $OUT/ab_sweep.log
done
