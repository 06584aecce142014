"""Follow-up to placement_candidates.py: (1) hipMalloc candidates FIRST, torch candidates after (is it the allocator or the order?);
(2) does a plain streaming read (ofmk_hbm_read) tell the fast from the slow allocations, or only the analyze kernel?  (3) the OUTPUT
buffer's placement; (4) free everything, allocate again: do the levels come back with the same blocks?
usage (GPU box): python tools/placement_candidates2.py [K=5]"""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, H, W = 300, 1080, 1920
nbytes = n * H * W * 3
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
torch.cuda.init()
torch.zeros(1, device="cuda")
raw = []
for i in range(K):                                  # hipMalloc candidates before anything else of size exists
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), nbytes) == 0
    raw.append(p.value)
eng = DctEngine()
lib = eng.lib
s = _hip.current_stream()
first = synthetic_frames(n, H, W, seed=2000)
outs = [torch.empty_like(first) for _ in range(3)]
ws = torch.empty(lib.ofmk_workspace_bytes(n, H, W), dtype=torch.uint8, device="cuda")
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda")
wm[0, ::2] = 1
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
pool = _hip.Timing(256, (1 << _hip.TIMING_KINDS.index("mark_fused")) | (1 << _hip.TIMING_KINDS.index("analyze")))


def run(ptr, out, k=8):
    o = _hip.Opts(0, 0, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(ptr, n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
        _hip.check(lib.ofmk_stage_mark_rgb8(ptr, out.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations()
    pool.collect()
    return (float(np.mean([x for x, kind in d if kind == "analyze"][2:])), float(np.mean([x for x, kind in d if kind == "mark_fused"][2:])))


def read_ms(ptr, k=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        _hip.check(lib.ofmk_hbm_read(ptr, nbytes, sink.data_ptr(), s))
    e0.record()
    for _ in range(k):
        _hip.check(lib.ofmk_hbm_read(ptr, nbytes, sink.data_ptr(), s))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


cands = [(f"hipMalloc#{i} (allocated first)", p) for i, p in enumerate(raw)]
for _, p in cands:
    assert hip.hipMemcpyAsync(p, C.c_void_p(first.data_ptr()), nbytes, 3, C.c_void_p(s)) == 0
keep = [first] + [first.clone() for _ in range(K - 1)]
cands += [(f"torch#{i} (allocated after)", t.data_ptr()) for i, t in enumerate(keep)]
torch.cuda.synchronize()
for _ in range(40):
    run(first.data_ptr(), outs[0], 4)
print("(1)+(2) candidate input allocations: analyze ms / fused mark ms / plain 16-B streaming read ms (3 interleaved rounds)")
acc = {name: [] for name, _ in cands}
for rnd in range(3):
    for name, p in cands:
        a, m = run(p, outs[0])
        acc[name].append((a, m, read_ms(p)))
for name, p in cands:
    v = np.mean(np.asarray(acc[name]), axis=0)
    print(f"  {name:34s} at {p / 2 ** 30:10.2f} GiB   analyze {v[0]:.4f}   fused mark {v[1]:.4f}   read {v[2]:.4f}")
best = min(cands, key=lambda c: np.mean([x[0] + x[1] for x in acc[c[0]]]))
worst = max(cands, key=lambda c: np.mean([x[0] + x[1] for x in acc[c[0]]]))
print(f"(3) output buffer placement, best input ({best[0]}) and worst input ({worst[0]}): fused mark ms per output buffer")
for name, p in (best, worst):
    print("  " + name + ": " + "  ".join(f"{run(p, o)[1]:.4f}" for o in outs))
print("(4) free every torch candidate (empty_cache), allocate K again")
addr_before = [t.data_ptr() for t in keep[1:]]
del keep[1:]
torch.cuda.empty_cache()
again = [first.clone() for _ in range(K - 1)]
torch.cuda.synchronize()
for i, t in enumerate(again):
    a, m = run(t.data_ptr(), outs[0])
    print(f"  new torch#{i + 1} at {t.data_ptr() / 2 ** 30:10.2f} GiB (before: {addr_before[i] / 2 ** 30:10.2f})   analyze {a:.4f}   fused mark {m:.4f}")
