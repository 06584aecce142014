"""Does the read speed of the INPUT frames depend on their virtual offset inside an allocation, or only on the allocation (tools/placement_experiment.py)?
300 x 1080p; K big allocations; in each, the frames are copied to byte offset `off` and analyze / fused mark are timed.
usage: python tools/input_offset_experiment.py [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark import _hip
from offmark.engine import DctEngine
from offmark.synthetic import synthetic_frames
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, H, W = 300, 1080, 1920
eng = DctEngine(tile_order="xcd")
lib = eng.lib
first = synthetic_frames(n, H, W, seed=2000)
size = first.numel()
bigs = [torch.empty(size + (8 << 20), dtype=torch.uint8, device="cuda") for _ in range(K)]
out = torch.empty_like(first)
ws = eng.workspace(H, W, n)
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda"); wm[0, ::2] = 1
s = _hip.current_stream()
pool = _hip.Timing(64, (1 << _hip.TIMING_KINDS.index("mark_fused")) | (1 << _hip.TIMING_KINDS.index("analyze")))
def run(src, k=10):
    o = _hip.Opts(0, 0, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(src.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
        _hip.check(lib.ofmk_stage_mark_rgb8(src.data_ptr(), out.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations(); pool.collect()
    return float(np.mean([x for x, kd in d if kd == "analyze"][2:])), float(np.mean([x for x, kd in d if kd == "mark_fused"][2:]))
for _ in range(30): run(first, 4)
print("the frames' own allocation: analyze %.4f  mark %.4f" % run(first))
offs = [0, 8, 64, 256, 1024, 4096, 65536, 1 << 20, (2 << 20) + 8, (4 << 20) + 4096 + 64]
for b, big in enumerate(bigs):
    row = []
    for off in offs:
        src = big[off:off + size].view(n, H, W, 3)
        src.copy_(first)
        a, m = run(src)
        row.append(f"{a:.4f}/{m:.4f}")
    print(f"allocation {b} at {big.data_ptr() / 2**30:.2f} GiB, analyze/mark by offset {offs}:\n   " + "  ".join(row))
