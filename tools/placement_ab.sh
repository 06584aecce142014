# interleaved A/B of bench.py with and without placement-probed workspace / output buffers (--placement-candidates 1 | 8), a fresh process each
for i in 1 2 3 4; do
  for k in 1 8; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --placement-candidates $k > gpurun_out/r6_place_${k}_$i.json 2> gpurun_out/r6_place_${k}_$i.err || echo "rc=$? k=$k i=$i"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_place_*_*.json")):
    l=json.load(open(f)); pp=l["config"]["placement_probe"]
    print(f.split("/")[-1], l["value"], "fused", l["roofline"]["avg_launch_ms"], "analyze", (l["kernels"].get("analyze") or {}).get("avg_launch_ms"), (pp.get("workspace") or {}), (pp.get("output") or {}))
PY
