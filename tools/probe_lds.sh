#!/bin/bash
# Is the MX GEMM's K loop bound by LDS read bytes?  AVL_MX_PROBE=4 drops the weight-fragment reads (1/3 of the LDS read bytes),
# 5 every other activation-fragment read (another 1/3); results are garbage, only the times matter.
. tools/use_experiments_lib.sh
OUT=${1:-gpurun_out/r3}
mkdir -p $OUT
for p in ${PROBES:-0 4 5 2}; do
  echo "=== AVL_MX_PROBE=$p"
  AVL_MX_PROBE=$p python tools/profile_seg.py --precision mixed --top 12 --reps 3 2>&1 | grep -v amdgpu.ids | grep "gemm  \|layer4\|layer3.1"
done | tee $OUT/probe_lds.log
