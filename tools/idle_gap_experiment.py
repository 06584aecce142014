"""How long must the device sit idle before the next launches run slower?  (the first launches of bench.py's timed region,
which the contract starts right after a barrier + synchronize, take 5-8 % longer than the same kernel a few milliseconds later.)
300 x 1080p; 150 ms of back-to-back analyze + fused mark launches, synchronize, sleep `gap`, then 12 timed analyze + mark pairs.
usage: python tools/idle_gap_experiment.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark import _hip
from offmark.engine import DctEngine
from offmark.synthetic import synthetic_frames
n, H, W = 300, 1080, 1920
eng = DctEngine(tile_order="xcd")
lib = eng.lib
src = synthetic_frames(n, H, W, seed=2000)
dst = torch.empty_like(src)
ws = eng.workspace(H, W, n)
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda"); wm[0, ::2] = 1
s = _hip.current_stream()
kind = 1 << _hip.TIMING_KINDS.index("mark_fused")
pool = _hip.Timing(64, kind)
o_t = _hip.Opts(0, 0, pool.handle)
def pair(timed):
    _hip.check(lib.ofmk_stage_analyze_rgb8(src.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, None))
    _hip.check(lib.ofmk_stage_mark_rgb8(src.data_ptr(), dst.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s,
                                        _hip.opts_ref(o_t) if timed else None))
def busy(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8): pair(False)
        torch.cuda.synchronize()
busy(400)
print("gap_ms | fused mark launch durations after the gap (ms)")
for rep in range(2):
    for gap in (0.0, 0.05, 0.2, 1.0, 5.0, 20.0, 100.0):
        busy(150)
        for _ in range(8): pair(False)
        torch.cuda.synchronize()
        if gap: 
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < gap: pass
        for _ in range(12): pair(True)
        torch.cuda.synchronize()
        d = [m for m, _ in pool.durations()]
        pool.collect()
        print(f"{gap:6.2f} | " + " ".join(f"{x:.3f}" for x in d))
# no synchronize at all between the busy phase and the timed launches
busy(150)
for _ in range(8): pair(False)
for _ in range(12): pair(True)
torch.cuda.synchronize()
print("  none | " + " ".join(f"{m:.3f}" for m, _ in pool.durations()))
