#!/bin/bash
# Build the library at git HEAD as gpurun_out-independent variant A (committed code) next to the working tree's B.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
TMP=$(mktemp -d)
git -C $ROOT archive HEAD vision_semantic_segmentation_amd/csrc include | tar -x -C $TMP
make -s -C $TMP/vision_semantic_segmentation_amd/csrc -j4 OUT=$ROOT/vision_semantic_segmentation_amd/libavl_hip_A.so ROOT=$TMP >/dev/null
make -s -C $ROOT/vision_semantic_segmentation_amd/csrc -j4 >/dev/null
cp $ROOT/vision_semantic_segmentation_amd/libavl_hip.so $ROOT/vision_semantic_segmentation_amd/libavl_hip_B.so
rm -rf $TMP
ls -la $ROOT/vision_semantic_segmentation_amd/*.so
