#!/bin/bash
# What bounds the software-pipelined MX GEMM?  AVL_MX_PROBE: 0 full kernel, 1 no DMA after the prologue (MFMA + LDS reads + barriers),
# 2 no MFMAs (DMA + LDS reads + barriers).  Per-op profile of the mixed plan; layer4 conv1 / conv3 are the shapes to read.
OUT=${1:-gpurun_out/r3}
mkdir -p $OUT
for p in 0 1 2; do
  echo "=== AVL_MX_PROBE=$p" | tee -a $OUT/probe_mx.log
  AVL_MX_PROBE=$p python tools/profile_seg.py --precision mixed --top 14 --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/probe_mx.log | grep -E "^gemm|layer4.1.conv1|layer4.1.conv3|layer3.2.conv3 |layer3.2.conv1 "
done
