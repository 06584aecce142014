"""How much do CANDIDATE allocations of one process differ?  K input buffers (same bytes; all alive, so all at different physical
places), allocated three ways -- torch's caching allocator, hipMalloc through ctypes (no torch pool), and torch again after the
first set was freed --, each timed with analyze and the fused mark kernel into one common output buffer; rounds interleaved.
usage (GPU box): python tools/placement_candidates.py [K=6] [frames=300]"""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
H, W = 1080, 1920
eng = DctEngine()
lib = eng.lib
s = _hip.current_stream()
first = synthetic_frames(n, H, W, seed=2000)
nbytes = first.numel()
out = torch.empty_like(first)
ws = torch.empty(lib.ofmk_workspace_bytes(n, H, W), dtype=torch.uint8, device="cuda")
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda")
wm[0, ::2] = 1
pool = _hip.Timing(256, (1 << _hip.TIMING_KINDS.index("mark_fused")) | (1 << _hip.TIMING_KINDS.index("analyze")))


def run(ptr, k=8):
    o = _hip.Opts(0, 0, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(ptr, n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
        _hip.check(lib.ofmk_stage_mark_rgb8(ptr, out.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations()
    pool.collect()
    return (float(np.mean([x for x, kind in d if kind == "analyze"][2:])), float(np.mean([x for x, kind in d if kind == "mark_fused"][2:])))


hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
cands = [("torch#0 (the frames' own allocation)", first.data_ptr(), first)]
for i in range(1, K):
    t = first.clone()
    cands.append((f"torch#{i}", t.data_ptr(), t))
for i in range(K):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), nbytes) == 0
    assert hip.hipMemcpyAsync(p, C.c_void_p(first.data_ptr()), nbytes, 3, C.c_void_p(s)) == 0
    cands.append((f"hipMalloc#{i}", p.value, None))
torch.cuda.synchronize()
for _ in range(40):
    run(first.data_ptr(), 4)
res = {name: [] for name, _, _ in cands}
for rnd in range(3):
    for name, ptr, _ in cands:
        res[name].append(run(ptr))
print(f"{n} x 1080p, analyze ms / fused mark ms per candidate input allocation (3 interleaved rounds, mean)")
rows = []
for name, ptr, _ in cands:
    a = float(np.mean([x[0] for x in res[name]]))
    m = float(np.mean([x[1] for x in res[name]]))
    rows.append((a + m, name, ptr, a, m))
    print(f"  {name:40s} at {ptr / 2 ** 30:10.2f} GiB   analyze {a:.4f}   fused mark {m:.4f}   sum {a + m:.4f}")
best, worst = min(rows), max(rows)
print(f"best {best[1]} {best[0]:.4f} ms, worst {worst[1]} {worst[0]:.4f} ms: spread {100 * (worst[0] / best[0] - 1):.1f} % of analyze + fused mark")
