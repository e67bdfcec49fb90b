mkdir -p gpurun_out/r2
rm -f gpurun_out/r2/gc_exp.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "grouped or gconv or logits" > gpurun_out/r2/gc_tests.log 2>&1 || { tail -30 gpurun_out/r2/gc_tests.log; exit 1; }
tail -3 gpurun_out/r2/gc_tests.log
for e in 0 4; do
  echo "== AVL_GCONV_TH=$e" >> gpurun_out/r2/gc_exp.log
  AVL_GCONV_TH=$e python tools/profile_seg.py --precision mixed --kind gconv --reps 3 2>&1 | grep -v "^$" >> gpurun_out/r2/gc_exp.log || exit 1
done
