"""Does the fused mark kernel's duration depend on where the output buffer sits relative to the input buffer?  (Two bench.py processes
on one box differ by up to 5 % in this kernel's duration with identical code; both are hipMalloc'ed at 2 MiB granularity.)
300 x 1080p; out = a slice of one big buffer at byte offset `off`; 12 analyze + fused mark pairs per offset, mean of the last 10, 3 rounds.
usage: python tools/offset_experiment.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark import _hip
from offmark.engine import DctEngine
from offmark.synthetic import synthetic_frames
n, H, W = 300, 1080, 1920
size = n * H * W * 3
eng = DctEngine(tile_order="xcd")
lib = eng.lib
src = synthetic_frames(n, H, W, seed=2000)
big = torch.empty(size + (64 << 20), dtype=torch.uint8, device="cuda")
ws = eng.workspace(H, W, n)
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda"); wm[0, ::2] = 1
s = _hip.current_stream()
pool = _hip.Timing(64, 1 << _hip.TIMING_KINDS.index("mark_fused"))
def run(dst, flags, k=12):
    o = _hip.Opts(flags, 0, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(src.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, None))
        _hip.check(lib.ofmk_stage_mark_rgb8(src.data_ptr(), dst.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = [m for m, _ in pool.durations()][2:]
    pool.collect()
    return float(np.mean(d))
for _ in range(40): run(big[:size].view(n, H, W, 3), 0, 4)          # bring the device to its operating state
print(f"in at {src.data_ptr():#x}, big at {big.data_ptr():#x}, delta {(big.data_ptr() - src.data_ptr()) / 2**20:.1f} MiB")
offs = [0, 256, 1024, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, (1 << 21), (1 << 21) + 4096, (1 << 22) + 65536 + 256, 33 << 20]
print("offset    xcd_ms  linear_ms   (3 rounds)")
res = {o: [] for o in offs}
for rnd in range(3):
    for off in offs:
        dst = big[off:off + size].view(n, H, W, 3)
        res[off].append((run(dst, 0), run(dst, _hip.F_LINEAR_TILES)))
for off in offs:
    print(f"{off:10d}  " + "  ".join(f"{a:.4f}/{b:.4f}" for a, b in res[off]))
# in-place for comparison
print("in place  ", "  ".join(f"{run(src, 0):.4f}" for _ in range(3)))
