#!/bin/bash
# Logits error and frame time of one mixed-mode builder option against the default, same box:
#   bash tools/opt_check.sh layer1_lo=0      (tools/seed_sweep.py cases + tools/profile_seg.py)
OPT=${1:?option, e.g. layer1_lo=0}
OUT=${2:-gpurun_out/r3}
mkdir -p $OUT
( for o in - $OPT; do
    echo "=== mixed options: $o"
    SWEEP_SHORT=1 python tools/seed_sweep.py $o 2>&1 | grep -v amdgpu.ids | grep "weights seed [02]"
    python tools/profile_seg.py --precision mixed --top 3 --reps 5 $( [ "$o" != "-" ] && echo --mixed-opts $o ) 2>&1 | grep -E "^total|plan run"
  done ) | tee $OUT/opt_check_${OPT%%=*}.log
