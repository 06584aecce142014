"""Which buffer's placement moves the fused mark kernel?  One process, 300 x 1080p: K different hipMalloc'ed output buffers (all kept alive, so all
at different physical places), K different input buffers (same contents), K different workspaces; every (in, out, ws) triple timed.
usage: python tools/placement_experiment.py [K]"""
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
outs = [torch.empty_like(first) for _ in range(K)]
nb = lib.ofmk_workspace_bytes(n, H, W)
wss = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(K)]
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda"); wm[0, ::2] = 1
s = _hip.current_stream()
pool = _hip.Timing(64, (1 << _hip.TIMING_KINDS.index("mark_fused")) | (1 << _hip.TIMING_KINDS.index("analyze")))
def run(src, dst, ws, flags, k=10):
    o = _hip.Opts(flags, 0, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(src.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
        _hip.check(lib.ofmk_stage_mark_rgb8(src.data_ptr(), dst.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations()
    pool.collect()
    m = [x for x, kind in d if kind == "mark_fused"][2:]
    a = [x for x, kind in d if kind == "analyze"][2:]
    return float(np.mean(m)), float(np.mean(a))
for _ in range(30): run(ins[0], outs[0], wss[0], 0, 4)
print("addresses (GiB):", "in", [round(t.data_ptr() / 2**30, 2) for t in ins], "out", [round(t.data_ptr() / 2**30, 2) for t in outs])
for rnd in range(2):
    print(f"round {rnd}: rows = input buffer, columns = output buffer; fused mark ms xcd/linear (analyze ms of that input in brackets)")
    for i in range(K):
        cells = []
        for j in range(K):
            mx, a = run(ins[i], outs[j], wss[0], 0)
            ml, _ = run(ins[i], outs[j], wss[0], _hip.F_LINEAR_TILES)
            cells.append(f"{mx:.4f}/{ml:.4f}")
        print(f"  in{i} [{a:.4f}]  " + "  ".join(cells))
print("workspace placement (in0, out0): " + "  ".join(f"{run(ins[0], outs[0], w, 0)[0]:.4f}" for w in wss))
