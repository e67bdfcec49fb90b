mkdir -p gpurun_out/r2
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "grouped or gconv or logits" > gpurun_out/r2/gc_tests.log 2>&1 || { tail -30 gpurun_out/r2/gc_tests.log; exit 1; }
tail -2 gpurun_out/r2/gc_tests.log
git stash -q 2>/dev/null
python tools/profile_seg.py --precision mixed --kind gconv --reps 3 > gpurun_out/r2/gc_new.log 2>&1
echo "new: $(grep '^gconv' gpurun_out/r2/gc_new.log) $(grep 'plan run' gpurun_out/r2/gc_new.log)"
