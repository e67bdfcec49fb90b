#!/bin/bash
# A/B of two builds of the library on one box: per-op profile of the mixed plan with AVL_HIP_LIB = $BASE (default
# vision_semantic_segmentation_amd/libavl_hip_base.so, e.g. the previous commit's build) and with the current libavl_hip.so, interleaved.
OUT=${1:-gpurun_out/r4}
BASE=${BASE:-$PWD/vision_semantic_segmentation_amd/libavl_hip_base.so}
PAT=${PAT:-"^total|^gemm|^gconv|plan run|layer4.1.conv1|layer4.1.conv3|layer4.0.conv3|layer3.1.conv3|layer3.0.conv3|layer2.1.conv3"}
mkdir -p $OUT
for rep in 1 2; do
  for lib in $BASE $PWD/vision_semantic_segmentation_amd/libavl_hip.so; do
    echo "=== $(basename $lib)" | tee -a $OUT/ab_lib.log
    AVL_HIP_LIB=$lib timeout -k 10 200 python tools/profile_seg.py --precision mixed --top 100 --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_lib.log | grep -E "$PAT"
  done
done
