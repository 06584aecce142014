import sys, time, os
sys.path.insert(0, 'video-fingerprinting_amd')
import numpy as np
from offmark.video import pipeline as pl
print('cores', len(os.sched_getaffinity(0)), 'copy threads', pl._COPY_THREADS)
src = np.random.default_rng(0).integers(0, 256, (50, 1080, 1920, 3), dtype=np.uint8)
pin = pl.pinned_empty(src.shape)
for thr in (1, 2, 4, 8, 16):
    pl._COPY_THREADS = thr; pl._pool = None
    pl.host_copy(pin, src)
    t0 = time.perf_counter()
    for _ in range(5): pl.host_copy(pin, src)
    dt = (time.perf_counter() - t0) / 5
    fresh = []
    t1 = time.perf_counter()
    for _ in range(3):
        d = np.empty_like(src); pl.host_copy(d, pin); fresh.append(d)
    dt2 = (time.perf_counter() - t1) / 3
    print(f"threads {thr:2d}: pageable->pinned {src.nbytes / dt / 1e9:6.1f} GB/s   pinned->fresh pageable {src.nbytes / dt2 / 1e9:6.1f} GB/s")
    del fresh
