# Sourced by the tools/ scripts that drive experiment switches (AVL_MX_PROBE, AVL_GEMM_PROBE, AVL_SWEEP_EXP, ...): those exist only in
# the EXPERIMENTS build of the library (`make -C vision_semantic_segmentation_amd/csrc experiments` -> libavl_hip_exp.so, built in
# this container so that it travels to the GPU box); the release libavl_hip.so reads no environment variable.
AVL_HIP_LIB=${AVL_HIP_LIB:-$PWD/vision_semantic_segmentation_amd/libavl_hip_exp.so}
if [ ! -f "$AVL_HIP_LIB" ]; then echo "missing $AVL_HIP_LIB: run 'make -C vision_semantic_segmentation_amd/csrc experiments' first" >&2; exit 2; fi
export AVL_HIP_LIB
