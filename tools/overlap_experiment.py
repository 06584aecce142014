"""Experiment (not product code): CAN this chip run the engine's arithmetic and its memory stream at full speed at the same
time?  Two kernels on two streams: V = analyze's real instruction stream with its pixel loads replaced by a register
initialisation (a library built with -DOFMK_EXPERIMENT_NO_LOADS=1: arithmetic + record stores only) and M = the read-only
streaming probe over the same 1.87 GB of frames.  If the pair finishes in about max(V, M) the hardware overlaps them and the
kernels' structure is what loses the overlap; if it takes about V + M (or each slows down) the limit is shared (power, issue).
usage: python tools/overlap_experiment.py tools/bin/lib_noloads.so"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import torch
from offmark import _hip
from offmark.synthetic import synthetic_frames

lib = _hip.load()
nol = C.CDLL(os.path.abspath(sys.argv[1]))
for name in ("ofmk_stage_analyze_rgb8",):
    fn = getattr(nol, name)
    fn.restype, fn.argtypes = _hip.SIGNATURES[name]
n, H, W = 300, 1080, 1920
frames = synthetic_frames(n, H, W, seed=2000)
nbytes = frames.numel() // 16 * 16
ws = torch.empty(lib.ofmk_workspace_bytes(n, H, W), dtype=torch.uint8, device="cuda")
ws2 = torch.empty_like(ws)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def V(stream, reps):
    for _ in range(reps):
        _hip.check(nol.ofmk_stage_analyze_rgb8(frames.data_ptr(), n, H, W, ws.data_ptr(), ws.numel(), stream.cuda_stream, None))


def M(stream, reps):
    for _ in range(reps):
        _hip.check(lib.ofmk_hbm_read(frames.data_ptr(), nbytes, sink.data_ptr(), stream.cuda_stream))


def A(stream, reps):          # the shipped analyze: arithmetic and loads in the same waves
    for _ in range(reps):
        _hip.check(lib.ofmk_stage_analyze_rgb8(frames.data_ptr(), n, H, W, ws2.data_ptr(), ws2.numel(), stream.cuda_stream, None))


def wall(fn, reps=20):
    fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(reps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


v = wall(lambda r: V(s1, r))
m = wall(lambda r: M(s2, r))
a = wall(lambda r: A(s1, r))


def both(r):
    V(s1, r)
    M(s2, r)


vm = wall(both)
print(f"V alone (analyze's arithmetic, no pixel loads): {v:.4f} ms   M alone (read-only stream of the same frames): {m:.4f} ms")
print(f"V and M concurrently on two streams: {vm:.4f} ms per pair   (max = {max(v, m):.4f}, sum = {v + m:.4f})")
print(f"shipped analyze (same arithmetic + the loads, one kernel): {a:.4f} ms")
