"""Does the WORKSPACE's placement (the records: 12 B per block written by analyze, read and re-written by the fused mark kernel) decide the level a
process runs at?  One input batch, one output buffer, K candidate workspaces (all alive at once); analyze and fused mark timed per workspace,
interleaved rounds.  Background: tools/probe_ladder.hip's analyze rungs -- the record stores cost analyze 0.05-0.07 ms although they are 6 % of its bytes,
and the fast / slow difference between boxes sits entirely in that part (profiles/r6_mark_ladder.txt).
usage (GPU box): python tools/placement_workspace.py [K=8] [frames=300]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from offmark import _hip  # noqa: E402
from offmark.engine import DctEngine  # noqa: E402
from offmark.synthetic import synthetic_frames  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
H, W = 1080, 1920
eng = DctEngine()
lib = eng.lib
s = _hip.current_stream()
frames = synthetic_frames(n, H, W, seed=2000)
out = torch.empty_like(frames)
nb = lib.ofmk_workspace_bytes(n, H, W)
wss = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(K)]
spacer = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")            # candidates allocated after a gap, too
wss += [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(K // 2)]
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda")
wm[0, ::2] = 1
pool = _hip.Timing(256, (1 << _hip.TIMING_KINDS.index("mark_fused")) | (1 << _hip.TIMING_KINDS.index("analyze")))


def run(ws, k=8):
    o = _hip.Opts(0, 0, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(frames.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
        _hip.check(lib.ofmk_stage_mark_rgb8(frames.data_ptr(), out.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations()
    pool.collect()
    return (float(np.mean([x for x, kind in d if kind == "analyze"][2:])), float(np.mean([x for x, kind in d if kind == "mark_fused"][2:])))


for _ in range(40):
    run(wss[0], 4)
acc = [[] for _ in wss]
for rnd in range(4):
    order = range(len(wss)) if rnd % 2 == 0 else range(len(wss) - 1, -1, -1)
    for i in order:
        acc[i].append(run(wss[i]))
print(f"{n} x 1080p, ONE input (at {frames.data_ptr() / 2 ** 30:.2f} GiB) and output; analyze ms / fused mark ms per candidate WORKSPACE (4 interleaved rounds, median)")
tot = []
for i, w in enumerate(wss):
    v = np.median(np.asarray(acc[i]), axis=0)
    tot.append(v[0] + v[1])
    print(f"  workspace#{i:<2d} at {w.data_ptr() / 2 ** 30:10.2f} GiB   analyze {v[0]:.4f}   fused mark {v[1]:.4f}   sum {v[0] + v[1]:.4f}")
print(f"best {min(tot):.4f} ms, worst {max(tot):.4f} ms: spread {100 * (max(tot) / min(tot) - 1):.1f} %")
