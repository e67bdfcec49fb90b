#!/bin/bash
# Same-box A/B of seg_bottleneck.hip tuning switches: `bash tools/ab_bottleneck.sh build 0 1 2 ...` builds libavl_hip_bn<N>.so for every BN_EXP
# value given (in this container, before gpurun), then `bash tools/ab_bottleneck.sh run 0 1 2 ...` on the GPU box times
# tools/bench_bottleneck.py with each library, interleaved twice.
set -e
CS=vision_semantic_segmentation_amd/csrc
if [ "$1" = "build" ]; then
  shift
  for v in "$@"; do
    make -s -C $CS -j6 OUT=$PWD/vision_semantic_segmentation_amd/libavl_hip_bn$v.so BUILD=build_bn$v EXPFLAGS=-DBN_EXP=$v
  done
  exit 0
fi
shift
for rep in 1 2; do
  for v in "$@"; do
    echo "=== BN_EXP=$v (run $rep)"
    AVL_HIP_LIB=$PWD/vision_semantic_segmentation_amd/libavl_hip_bn$v.so python tools/bench_bottleneck.py 2>&1 | grep layer1
  done
done
