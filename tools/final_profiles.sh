set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/final gpurun_out/r5
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
bash tools/prof_bench.sh > gpurun_out/final/prof_bench.out 2>&1
bash tools/pmc_seg.sh mixed > gpurun_out/final/pmc_seg.out 2>&1
bash tools/prof_mapping.sh > gpurun_out/final/prof_mapping.out 2>&1
python tools/profile_seg.py --precision mixed --reps 3 --top 100 > gpurun_out/final/profile_seg_mixed_per_op.log 2>&1
tail -c 300 gpurun_out/final/bench.json
