#!/bin/bash
# Logits error of the DEFAULT mixed mode against the fp32 oracle over weight seeds x frames (tools/seed_sweep.py), round 3.
OUT=${1:-gpurun_out/r3}
mkdir -p $OUT
python tools/seed_sweep.py - 2>&1 | grep -v amdgpu.ids | tee $OUT/seed_sweep.log
python tools/seed_sweep.py - wide 2>&1 | grep -v amdgpu.ids | tee -a $OUT/seed_sweep.log
