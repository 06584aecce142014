"""Does physically contiguous VRAM (hipExtMallocWithFlags + hipDeviceMallocContiguous) remove the placement lottery of the frame buffers?
One process, 300 x 1080p: K default allocations (torch / hipMalloc) and K contiguous ones for the INPUT frames (the buffer whose placement
matters, tools/placement_experiment.py), same contents; analyze and fused mark timed on each.
usage: python tools/contiguous_experiment.py [K]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark import _hip
from offmark.engine import DctEngine
from offmark.synthetic import synthetic_frames
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, H, W = 300, 1080, 1920
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
class Ext:
    def __init__(self, nbytes, flags):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flags)
        if rc != 0 or not p.value:
            raise MemoryError(f"hipExtMallocWithFlags({nbytes}, {flags}) -> {rc}")
        self.ptr, self.nbytes = p.value, nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (p.value, False), "version": 2}
    def tensor(self, shape):
        return torch.as_tensor(self, device="cuda").view(shape)
eng = DctEngine(tile_order="xcd")
lib = eng.lib
first = synthetic_frames(n, H, W, seed=2000)
size = first.numel()
bufs = [("default", first)] + [("default", first.clone()) for _ in range(K - 1)]
keep = []
for i in range(K):
    try:
        e = Ext(size, 0x4)                      # hipDeviceMallocContiguous
        keep.append(e)
        t = e.tensor(first.shape)
        t.copy_(first)
        bufs.append(("contiguous", t))
    except Exception as exc:
        print("contiguous allocation failed:", exc)
out = torch.empty_like(first)
try:
    eo = Ext(size, 0x4); keep.append(eo); out_c = eo.tensor(first.shape)
except Exception as exc:
    out_c = None; print("contiguous out failed:", exc)
ws = eng.workspace(H, W, n)
wm = torch.zeros((1, H * W // 64), dtype=torch.uint8, device="cuda"); wm[0, ::2] = 1
s = _hip.current_stream()
pool = _hip.Timing(64, (1 << _hip.TIMING_KINDS.index("mark_fused")) | (1 << _hip.TIMING_KINDS.index("analyze")))
def run(src, dst, flags=0, k=10):
    o = _hip.Opts(flags, 0, pool.handle)
    for _ in range(k):
        _hip.check(lib.ofmk_stage_analyze_rgb8(src.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
        _hip.check(lib.ofmk_stage_mark_rgb8(src.data_ptr(), dst.data_ptr(), n, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, _hip.opts_ref(o)))
    torch.cuda.synchronize()
    d = pool.durations(); pool.collect()
    return float(np.mean([x for x, kd in d if kd == "mark_fused"][2:])), float(np.mean([x for x, kd in d if kd == "analyze"][2:]))
for _ in range(30): run(first, out, 0, 4)
ref = None
for rnd in range(2):
    print(f"round {rnd}: input allocation -> analyze ms | fused mark ms (xcd) | fused mark ms (linear)" + (" | mark into a contiguous output" if out_c is not None else ""))
    for kind, b in bufs:
        mx, a = run(b, out, 0)
        ml, _ = run(b, out, _hip.F_LINEAR_TILES)
        extra = f" | {run(b, out_c, 0)[0]:.4f}" if out_c is not None else ""
        print(f"  {kind:10s} at {b.data_ptr() / 2**30:10.2f} GiB: {a:.4f} | {mx:.4f} | {ml:.4f}{extra}")
res = torch.equal(eng.embed(bufs[0][1], wm), eng.embed(bufs[-1][1], wm))
print("same results from both kinds of buffer:", res)
