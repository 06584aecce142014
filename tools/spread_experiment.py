"""Does spreading the concurrently active tiles over the whole buffer (tile order over X pseudo-XCDs, X = ofmk_opts.xcds up to 64) level out the
allocation-dependent read speeds (tools/placement_experiment.py)?  K input allocations, fused mark kernel, X in {linear, 8, 16, 24, 32, 64}.
usage: python tools/spread_experiment.py [K]"""
import os, sys
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
out = torch.empty_like(first)
ws = eng.workspace(H, W, n)
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda"); wm[0, ::2] = 1
s = _hip.current_stream()
pool = _hip.Timing(64, (1 << _hip.TIMING_KINDS.index("mark_fused")) | (1 << _hip.TIMING_KINDS.index("analyze")))
def run(src, flags, xcds, k=10):
    o = _hip.Opts(flags, xcds, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(src.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
        _hip.check(lib.ofmk_stage_mark_rgb8(src.data_ptr(), out.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations(); pool.collect()
    return float(np.mean([x for x, kd in d if kd == "mark_fused"][2:])), float(np.mean([x for x, kd in d if kd == "analyze"][2:]))
for _ in range(30): run(ins[0], 0, 0, 4)
variants = [("linear", _hip.F_LINEAR_TILES, 0), ("X=8", 0, 8), ("X=16", 0, 16), ("X=24", 0, 24), ("X=32", 0, 32), ("X=64", 0, 64)]
print("fused mark ms by tile order (analyze ms of the same input, linear order, in brackets)")
for rnd in range(2):
    for i, b in enumerate(ins):
        cells, a = [], 0
        for name, fl, x in variants:
            m, a = run(b, fl, x)
            cells.append(f"{name} {m:.4f}")
        print(f"  round {rnd} input {i} [{a:.4f}]: " + "   ".join(cells))
