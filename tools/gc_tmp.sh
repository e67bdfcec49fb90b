mkdir -p gpurun_out/r2
timeout -k 10 500 python -m pytest tests/test_gpu_mixed.py -x -q > gpurun_out/r2/mx_tests.log 2>&1 || { tail -30 gpurun_out/r2/mx_tests.log; exit 1; }
tail -2 gpurun_out/r2/mx_tests.log
python tools/profile_seg.py --precision mixed --reps 3 --top 100 > gpurun_out/r2/prof_new.log 2>&1
echo "new: $(grep '^gemm' gpurun_out/r2/prof_new.log) $(grep 'plan run' gpurun_out/r2/prof_new.log)"
grep "layer4.1.conv3\|layer3.3.conv3\|layer4.1.conv1" gpurun_out/r2/prof_new.log
