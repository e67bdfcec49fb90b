#!/bin/bash
# Where does a wave of the software-pipelined MX GEMM spend its cycles?  AVL_MX_PROBE=3: s_memtime stamps around the two events and
# the two DMA bursts of every sub-step, per-wave sums printed by the launcher (256-row tiles only).  Each stamp drains the wave's
# LDS queue, so the build is slower than the real kernel: read the shares.
. tools/use_experiments_lib.sh
OUT=${1:-gpurun_out/r3}
mkdir -p $OUT
AVL_MX_PROBE=3 python tools/profile_seg.py --precision mixed --top 4 --reps 1 2>&1 | grep -v amdgpu.ids | grep "mx probe" | sort | uniq -c | sort -k4,4n -k6,6n -k8,8n | tee $OUT/probe_mx_stamps.log
