#!/bin/bash
# Sample engine clock / power with rocm-smi while bench.py runs (is the kernel mix power-limited?).
python bench.py --steps 30000 --warmup 3 --no-cpu-baseline --no-kernel-events > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err &
BP=$!
sleep 12
for i in $(seq 1 14); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower --showuse 2>&1 | grep -v "^=\|^$" >> gpurun_out/clk.log
  echo "--" >> gpurun_out/clk.log
  sleep 2.5
done
wait $BP
echo "idle:" >> gpurun_out/clk.log
rocm-smi --showclocks --showpower 2>&1 | grep -v "^=\|^$" >> gpurun_out/clk.log
