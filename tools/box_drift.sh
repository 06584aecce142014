#!/bin/bash
# Does a box get slower from process to process?  N fresh bench.py processes back to back, junction temperature / power / clock sampled between them.
# usage (GPU box): bash tools/box_drift.sh [N=8]
N=${1:-8}
for i in $(seq 1 $N); do
  python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/drift_$i.json 2> gpurun_out/drift_$i.err
  T=$(rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "junction|Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' ')
  python - <<PY
import json
l=json.load(open("gpurun_out/drift_$i.json"))
k=l["kernels"]
print("run $i: value", l["value"], "fused mark ms", l["roofline"]["avg_launch_ms"], "analyze ms", k["analyze"]["avg_launch_ms"], "| after the run: $T")
PY
done
