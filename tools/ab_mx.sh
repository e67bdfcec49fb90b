#!/bin/bash
# A/B of the MX GEMM kernels on one box: per-op profile of the mixed plan with the round-2 kernel (AVL_MX_PIPE=0) and the
# software-pipelined one (AVL_MX_PIPE=1; AVL_MX_LATE picks the step at which waves 0-3 issue their DMAs).
. tools/use_experiments_lib.sh
set -o pipefail
OUT=${1:-gpurun_out/r3}
mkdir -p $OUT
for cfg in "0 -1" "1 -1" "1 2" "1 -1"; do
  set -- $cfg
  echo "=== AVL_MX_PIPE=$1 AVL_MX_LATE=$2" | tee -a $OUT/ab_mx.log
  AVL_MX_PIPE=$1 AVL_MX_LATE=$2 python tools/profile_seg.py --precision mixed --top 12 ${MIXED_OPTS:+--mixed-opts $MIXED_OPTS} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_mx.log | grep -E "^total|^gemm|plan run|hipGraph|layer4.1.conv1|layer4.1.conv3|layer3.2.conv3 "
done
