#!/bin/bash
# Samples GPU clock and power (rocm-smi) while bench.py runs: is the frame loop clock/power-limited?
#   gpurun -- 'bash tools/power_trace.sh'  ->  gpurun_out/power/{smi.log,bench.json}
OUT=gpurun_out/power
rm -rf $OUT; mkdir -p $OUT
rocm-smi --showclocks --showpower > $OUT/idle.log 2>&1
python3 bench.py --steps 600 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
for i in $(seq 1 60); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "== t=$i" >> $OUT/smi.log
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk\|fclk" >> $OUT/smi.log
  sleep 1
done
wait $BP
echo "bench rc=$?"
grep -i "sclk\|Power" $OUT/smi.log | sort | uniq -c | sort -rn | head -20
