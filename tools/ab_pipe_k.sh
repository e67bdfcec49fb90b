#!/bin/bash
# Which MX main loop for the short-K shapes?  AVL_MX_PIPE=-1 (per-shape choice: pipe where K >= 1024), 1 (pipe everywhere), 0 (ring everywhere).
. tools/use_experiments_lib.sh
OUT=${1:-gpurun_out/r4}
mkdir -p $OUT
for v in -1 1 0 -1 1; do
  echo "=== AVL_MX_PIPE=$v" | tee -a $OUT/ab_pipe_k.log
  AVL_MX_PIPE=$v timeout -k 10 200 python tools/profile_seg.py --precision mixed --top 100 --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_pipe_k.log | grep -E "^total|^gemm  |plan run|layer3.1.conv3|layer3.0.conv3|layer2.1.conv3|layer2.0.conv3|refine_layers.0.pointwise|refine_layers.1.pointwise|layer3.1.conv1 "
done
