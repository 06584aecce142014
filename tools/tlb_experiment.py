"""Are the read-speed levels of different allocations (tools/placement_experiment.py) address-translation levels?  K input buffers with the same 300 x 1080p
frames; 8 analyze launches on each, in order, each timed -- run it under `rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY
GRBM_GUI_ACTIVE` and feed the counter CSV to this script's second mode:  python tools/tlb_experiment.py [K]   |   python tools/tlb_experiment.py --csv <dir> K"""
import os, sys
if len(sys.argv) > 1 and sys.argv[1] == "--csv":
    import csv, glob, collections
    d, K = sys.argv[2], int(sys.argv[3])
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "analyze_kernel" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)[-8 * K:]                              # the last K groups of 8 launches (the warm-up precedes them)
    for i in range(K):
        g = [by[j] for j in ids[8 * i:8 * i + 8]]
        mean = {c: sum(x.get(c, 0.0) for x in g) / len(g) for c in g[0]}
        print(f"input {i}: " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(mean.items())))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark import _hip
from offmark.engine import DctEngine
from offmark.synthetic import synthetic_frames
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, H, W = 300, 1080, 1920
eng = DctEngine(tile_order="xcd")
lib = eng.lib
first = synthetic_frames(n, H, W, seed=2000)
ins = [first] + [first.clone() for _ in range(K - 1)]
ws = eng.workspace(H, W, n)
s = _hip.current_stream()
pool = _hip.Timing(64, 1 << _hip.TIMING_KINDS.index("analyze"))
o = _hip.Opts(0, 0, pool.handle)
def run(src, k):
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(src.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = [m for m, _ in pool.durations()]; pool.collect()
    return d
for _ in range(12): run(ins[0], 8)
for i, b in enumerate(ins):
    d = run(b, 8)
    print(f"input {i} at {b.data_ptr() / 2**30:.2f} GiB: analyze {np.mean(d[2:]):.4f} ms")
