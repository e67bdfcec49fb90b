#!/bin/bash
# A/B of the MX GEMM's DMA schedule on one box (experiments build): bursts at the two events (AVL_MX_SPREAD=0, the release
# schedule) against one DMA instruction per step (AVL_MX_SPREAD=1); parity of the spread kernel first (pipe-kernel tests).
. tools/use_experiments_lib.sh
set -o pipefail
OUT=${1:-gpurun_out/r4}
mkdir -p $OUT
AVL_MX_SPREAD=${SPREAD_TEST:-1} timeout -k 10 500 python -m pytest tests/test_gpu_mixed.py -m gpu -x -q -k "mx_gemm" > $OUT/pytest_spread.log 2>&1; tail -2 $OUT/pytest_spread.log
for sp in ${SPREADS:-0 1 0 1}; do
  echo "=== AVL_MX_SPREAD=$sp" | tee -a $OUT/ab_spread.log
  AVL_MX_SPREAD=$sp python tools/profile_seg.py --precision mixed --top 12 --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_spread.log | grep -E "^total|^gemm|plan run|layer4.1.conv1|layer4.1.conv3|layer4.0.conv3|layer3.1.conv1 "
done
