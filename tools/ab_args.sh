#!/bin/bash
# Interleaved A/B of bench.py argument sets in one GPU session: tools/ab_args.sh ROUNDS "args A" "args B" ...
R=$1; shift
for r in $(seq 1 $R); do for a in "$@"; do echo "== $a" >> gpurun_out/sweep.log; python bench.py --steps 20 --warmup 3 --no-cpu-baseline $a >> gpurun_out/sweep.log 2>>gpurun_out/sweep.err; done; done
