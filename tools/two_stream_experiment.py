"""Experiment (not product code): analyze(chunk k+1) on one stream WHILE mark+verify(chunk k) runs on another, chunks small
enough that mark's re-read of the frames is served by the 256 MiB Infinity Cache.  Eager launches (the host enqueues a
step faster than the GPU runs it).  Compared with the shipped single-stream path.
usage: python tools/two_stream_experiment.py [chunk ...]"""
import faulthandler
import os
faulthandler.enable()
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np
import torch
from offmark import _hip
from offmark.engine import DctEngine
from offmark.generator.shuffler import Shuffler
from offmark.synthetic import synthetic_frames

lib = _hip.load()
n, H, W = 300, 1080, 1920
frames = synthetic_frames(n, H, W, seed=2000)
out = torch.empty_like(frames)
wm = torch.from_numpy(Shuffler(key=0).generate_wm(np.array([0, 1, 1, 0, 0, 1, 0, 1]), (1, H * W // 64)).astype(np.uint8)).cuda()
eng = DctEngine()
ref, _, _ = eng.embed_detect(frames, wm, L=8)
torch.cuda.synchronize()


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"shipped path (one stream, chunk 300): {timeit(lambda: eng.embed_detect(frames, wm, L=8, out=out)):.4f} ms per 300 frames")
fs = H * W * 3
for chunk in [int(a) for a in sys.argv[1:]] or [10, 15, 20, 30, 50]:
    nbytes = lib.ofmk_workspace_bytes(chunk, H, W)
    ws = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    chunks = [(f0, min(chunk, n - f0)) for f0 in range(0, n, chunk)]

    def pattern():
        ev_a = [torch.cuda.Event() for _ in chunks]
        ev_m = [torch.cuda.Event() for _ in chunks]
        for k, (f0, cf) in enumerate(chunks):
            w = ws[k & 1]
            with torch.cuda.stream(s1):
                if k >= 2:
                    s1.wait_event(ev_m[k - 2])                       # the workspace half is free again
                _hip.check(lib.ofmk_stage_analyze_rgb8(frames.data_ptr() + f0 * fs, cf, H, W, w.data_ptr(), w.numel(), s1.cuda_stream, None))
                ev_a[k].record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(ev_a[k])
                _hip.check(lib.ofmk_stage_mark_rgb8(frames.data_ptr() + f0 * fs, out.data_ptr() + f0 * fs, cf, H, W, wm.data_ptr(), 20.0, 1,
                                                    w.data_ptr(), w.numel(), s2.cuda_stream, None))
                ev_m[k].record(s2)

    out.zero_()
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        s1.wait_stream(main); s2.wait_stream(main)
        pattern()
        main.wait_stream(s1); main.wait_stream(s2)
    torch.cuda.synchronize()
    ok = torch.equal(out, ref)
    def eager():
        with torch.cuda.stream(main):
            s1.wait_stream(main); s2.wait_stream(main)
            pattern()
            main.wait_stream(s1); main.wait_stream(s2)
    t_host0 = time.perf_counter()
    eager()
    host_ms = (time.perf_counter() - t_host0) * 1e3
    print(f"two streams, chunk {chunk:3d} ({len(chunks)} chunks, eager; host enqueue {host_ms:.2f} ms): {timeit(eager):.4f} ms per 300 frames "
          f"(analyze + mark+verify only; marked frames identical: {ok})")
