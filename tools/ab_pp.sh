#!/bin/bash
# A/B on one box (experiments build): the ping-pong MX GEMM k_gemm_mx_pp (AVL_MX_PP=1: every 256-row-tile MX GEMM) against the release kernels
# (k_gemm_mx_pipe / k_gemm_ring_mx, AVL_MX_PP=0); parity of the new kernel first (MX GEMM unit tests).
. tools/use_experiments_lib.sh
set -o pipefail
OUT=${1:-gpurun_out/r4}
mkdir -p $OUT
AVL_MX_PP=1 timeout -k 10 500 python -m pytest tests/test_gpu_mixed.py -m gpu -x -q -k "mx_gemm" > $OUT/pytest_pp.log 2>&1; tail -2 $OUT/pytest_pp.log
for sp in 0 1 0 1; do
  echo "=== AVL_MX_PP=$sp" | tee -a $OUT/ab_pp.log
  AVL_MX_PP=$sp timeout -k 10 200 python tools/profile_seg.py --precision mixed --top 12 --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_pp.log | grep -E "^total|^gemm|plan run|layer4.1.conv1|layer4.1.conv3|layer4.0.conv3|layer3.1.conv1 "
done
