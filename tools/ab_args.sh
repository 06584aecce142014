#!/bin/bash
# Interleaved A/B of bench.py argument sets in one GPU session: tools/ab_args.sh ROUNDS "args A" "args B" ...
# ($AB_COMMON: arguments every run gets; default: 20 steps after 3 warm-up steps, no CPU baseline)
R=$1; shift
COMMON=${AB_COMMON:---steps 20 --warmup 3 --no-cpu-baseline}
for r in $(seq 1 $R); do for a in "$@"; do echo "== $a" >> gpurun_out/sweep.log; python bench.py $COMMON $a >> gpurun_out/sweep.log 2>>gpurun_out/sweep.err; done; done
