#!/bin/bash
# Which pipes collide in the ring GEMM's K loop, and how much of it is the clock?  AVL_GEMM_PROBE: 0 full kernel, 1 DMA only,
# 2 LDS reads + MFMA (no DMA), 3 DMA + LDS reads (no MFMA); AVL_ZERO=1: all-zero activations (the chip holds a higher clock on
# trivial operands: MI355X_MICROARCH.md, DVFS give-back).  bf16 one-plane kernel on the network's GEMM shapes.
#   gpurun -- 'bash tools/gemm_probe.sh'
. tools/use_experiments_lib.sh
mkdir -p gpurun_out/probe
for z in "" 1; do
for p in 0 1 2 3; do
  echo "== AVL_GEMM_PROBE=$p AVL_ZERO=$z"
  AVL_ZERO=$z AVL_GEMM_PROBE=$p python tools/bench_gemm.py --variants 0 --no-check --reps 30 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe/p${p}_z${z}.log | head -3
done
done
