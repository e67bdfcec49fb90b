#!/bin/bash
# Same-box A/B of seg_dwpw.hip switches: `bash tools/ab_dwpw.sh build 0 1 2 ...` builds libavl_hip_dw<N>.so for every DW_EXP value given (in this
# container, before gpurun), then `bash tools/ab_dwpw.sh run 0 1 2 ...` on the GPU box times tools/bench_dwpw.py with each library.
set -e
CS=vision_semantic_segmentation_amd/csrc
if [ "$1" = "build" ]; then
  shift
  for v in "$@"; do
    # (a variant "N_M" also defines DW_IL=M)
    make -s -C $CS -j6 OUT=$PWD/vision_semantic_segmentation_amd/libavl_hip_dw$v.so BUILD=build_dw$v EXPFLAGS="-DDW_EXP=${v%%_*} -DDW_IL=$( [ "${v#*_}" != "$v" ] && echo ${v#*_} || echo 4 )"
  done
  exit 0
fi
shift
for v in "$@"; do
  echo "=== DW_EXP=$v"
  AVL_HIP_LIB=$PWD/vision_semantic_segmentation_amd/libavl_hip_dw$v.so python tools/bench_dwpw.py 2>&1 | grep -v "^$"
done
