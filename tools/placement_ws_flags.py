"""The workspace (records: written by analyze, read and re-written by the fused mark kernel, read by finalize) in memory allocated with
hipExtMallocWithFlags flags -- default, fine-grained, uncached -- against torch's allocations: analyze / fused mark / finalize-bearing detect ms.
usage (GPU box): python tools/placement_ws_flags.py [frames=300]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from offmark import _hip  # noqa: E402
from offmark.engine import DctEngine  # noqa: E402
from offmark.synthetic import synthetic_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, W = 1080, 1920
eng = DctEngine()
lib = eng.lib
s = _hip.current_stream()
frames = synthetic_frames(n, H, W, seed=2000)
out = torch.empty_like(frames)
nb = lib.ofmk_workspace_bytes(n, H, W)
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
cands = []
for i in range(3):
    t = torch.empty(nb, dtype=torch.uint8, device="cuda")
    cands.append((f"torch#{i}", t.data_ptr(), t))
for name, flag in (("default flag 0x0", 0x0), ("default flag 0x0 (2nd)", 0x0), ("fine-grained 0x1", 0x1), ("fine-grained 0x1 (2nd)", 0x1), ("uncached 0x3", 0x3), ("uncached 0x3 (2nd)", 0x3)):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nb, flag)
    if rc != 0:
        print(f"{name}: hipExtMallocWithFlags failed with {rc}")
        continue
    cands.append((name, p.value, None))
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda")
wm[0, ::2] = 1
counts = torch.empty((n, 8), dtype=torch.int32, device="cuda")
kinds = ("analyze", "mark_fused", "finalize")
pool = _hip.Timing(256, sum(1 << _hip.TIMING_KINDS.index(k) for k in kinds))


def run(ptr, k=6):
    o = _hip.Opts(0, 0, pool.handle)
    for _ in range(k):      # the real embed + detect step: analyze, fused mark + verify, finalize -- all three touch the records
        _hip.check(lib.ofmk_embed_detect_rgb8(frames.data_ptr(), out.data_ptr(), n, H, W, wm.data_ptr(), 1, None, 20.0, 8, counts.data_ptr(), None, n, ptr, nb, s,
                                              _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations()
    pool.collect()
    return [float(np.mean([x for x, kind in d if kind == kk][1:])) for kk in kinds]


for _ in range(30):
    run(cands[0][1], 4)
acc = {name: [] for name, _, _ in cands}
for rnd in range(4):
    order = cands if rnd % 2 == 0 else cands[::-1]
    for name, ptr, _ in order:
        acc[name].append(run(ptr))
print(f"{n} x 1080p embed+detect step, ms per kernel by where the WORKSPACE lives (4 interleaved rounds, median)")
for name, ptr, _ in cands:
    v = np.median(np.asarray(acc[name]), axis=0)
    print(f"  {name:26s} analyze {v[0]:.4f}   fused mark {v[1]:.4f}   finalize {v[2]:.4f}   sum {v.sum():.4f}")
