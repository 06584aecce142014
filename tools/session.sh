#!/bin/bash
# usage (through gpurun): bash tools/session.sh <tag> <step> [<step> ...]   -- runs the named measurement steps in order, each under its own
# timeout, writing to gpurun_out/<tag>_*; stops at the first step that is KILLED (timeout) so that no further GPU work follows a hang.
TAG=$1; shift
O=gpurun_out
mkdir -p $O
run() {   # run <seconds> <name> <command...>: stdout -> $O/${TAG}_<name>.json|log, stderr -> .err
    local secs=$1 name=$2; shift 2
    echo "== $name: $*"
    timeout -k 10 $secs "$@" > $O/${TAG}_$name.out 2> $O/${TAG}_$name.err
    local rc=$?
    echo "   rc=$rc"
    if [ $rc -ge 124 ]; then echo "step $name was killed: stopping the session"; exit $rc; fi
    return 0
}
B="python bench.py --no-cpu-baseline"
for step in "$@"; do
  case $step in
    tests)     AMD_LOG_LEVEL=1 run 1100 tests python -X faulthandler -m pytest tests -m gpu -v -x --capture=sys; tail -5 $O/${TAG}_tests.out; grep -v "^  File\|amdgpu.ids" $O/${TAG}_tests.err | head -20 ;;
    tests_all) AMD_LOG_LEVEL=1 run 1100 tests python -X faulthandler -m pytest tests -m gpu -v --capture=sys; tail -30 $O/${TAG}_tests.out | grep -v PASSED; grep -v "^  File\|amdgpu.ids" $O/${TAG}_tests.err | head -20 ;;
    memset)    run 200 graph_memset_order python tools/graph_memset_order.py; cat $O/${TAG}_graph_memset_order.out ;;
    line)      run 400 bench_line python bench.py ;;
    line20)    run 400 bench_line_steps20_warmup5 python bench.py --steps 20 --warmup 5 ;;
    c3)        run 600 bench_config3_4k_1000frames $B --config 3 --steps 20 --warmup 3 ;;
    c4)        run 300 bench_config4 $B --config 4 ;;
    c5)        run 300 bench_config5 $B --config 5 ;;
    c5svd)     run 300 bench_config5_dwtdctsvd $B --config 5 --codec dwtdctsvd ;;
    svd)       run 300 bench_line_dwtdctsvd $B --codec dwtdctsvd ;;
    svd8)      run 300 bench_line_dwtdctsvd_blk8 $B --codec dwtdctsvd --blk 8 ;;
    i420)      run 300 bench_line_i420 $B --pixfmt i420 ;;
    emu2)      run 300 emulate8_config2 $B --config 2 --emulate-world 8 ;;
    emu4)      run 300 emulate8_config4 $B --config 4 --emulate-world 8 ;;
    emu5)      run 300 emulate8_config5 $B --config 5 --emulate-world 8 ;;
    emu5svd)   run 300 emulate8_config5_dwtdctsvd $B --config 5 --emulate-world 8 --codec dwtdctsvd ;;
    emu4w2)    run 300 emulate2_config4 $B --config 4 --emulate-world 2 ;;
    emu4w4)    run 300 emulate4_config4 $B --config 4 --emulate-world 4 ;;
    sustained) run 300 bench_sustained_20000_steps $B --steps 20000 --warmup 5 --no-extras ;;
    prof)      run 900 prof bash tools/prof.sh ${TAG} ;;
    profsvd)   run 900 profsvd bash tools/prof.sh ${TAG}svd --codec dwtdctsvd ;;
    profsvd8)  run 900 profsvd8 bash tools/prof.sh ${TAG}svd8 --codec dwtdctsvd --blk 8 ;;
    profplanar) run 900 profplanar bash tools/prof.sh ${TAG}planar --pixfmt i420 ;;
    profembed) run 900 profembed bash tools/prof.sh ${TAG}embed --separate-detect ;;      # the NON-fused mark kernel (what tests/mark.py's operation runs) in the trace and the counters
    ladder)    run 1100 ladder bash tools/ladder_pmc.sh ${TAG} ;;
    *) echo "unknown step $step" ;;
  esac
done
for f in $O/${TAG}_bench_*.out $O/${TAG}_emulate*.out; do [ -s "$f" ] && mv "$f" "${f%.out}.json"; done
python - <<PY
import glob, json
for f in sorted(glob.glob("$O/${TAG}_*.json")):
    try:
        l = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" not in l: continue
    r = l.get("roofline") or {}
    print(f.split("/")[-1], "| value", l["value"], "| ms/step", l["ms_per_step"], "| second", l.get("value_second_pass"), "| dom", r.get("kernel"), r.get("avg_launch_ms"), r.get("frac"),
          "| ok", l["payload_bit_exact"], "| order", l["config"].get("tile_order"), "| host", l["host_ms_per_step"])
    if "emulation" in l:
        e = l["emulation"]; print("    emulation:", {k: e[k] for k in ("world", "shard_frames", "steps_per_host_iteration", "hipgraph", "shard_ms_per_step", "full_job_ms_per_step", "predicted_speedup", "predicted_frames_per_s", "host_ms_per_step")})
    if "mark_order" in l:
        m = l["mark_order"]; print("    mark_order:", {k: m.get(k) for k in ("xcd_ms", "linear_ms", "xcd_step_ms", "linear_step_ms", "shipped", "policy")})
    for k in ("embed_only", "detect_only"):
        if k in l and "value" in l[k]: print("    %s:" % k, {q: l[k].get(q) for q in ("value", "ms_per_step", "frac_of_peak", "frac_of_measured_copy", "frac_of_measured_read")})
PY
