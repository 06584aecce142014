"""Does marking in place (out == in) run faster than into a second buffer?  Same kernels; the write stream then lands on the
DRAM pages the read stream has just opened.  python tools/inplace_experiment.py   (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-fingerprinting_amd")]
from offmark.engine import DctEngine                     # noqa: E402
from offmark.generator.shuffler import Shuffler          # noqa: E402
from offmark.synthetic import synthetic_frames           # noqa: E402

n, H, W = 300, 1080, 1920
eng = DctEngine()
frames = synthetic_frames(n, H, W, seed=2000)
wm = torch.from_numpy(np.asarray(Shuffler(key=0).generate_wm(np.array([0, 1, 1, 0, 0, 1, 0, 1]), (1, H * W // 64)))[0].astype(np.uint8)[None]).cuda()
other = torch.empty_like(frames)


def run(out, steps=60, svd=False):
    for _ in range(5):
        (eng.svd_embed_detect(frames, wm, 8, out=out) if svd else eng.embed_detect(frames, wm, L=8, out=out))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        (eng.svd_embed_detect(frames, wm, 8, out=out) if svd else eng.embed_detect(frames, wm, L=8, out=out))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for rnd in range(3):
    print("round %d: DCT embed+detect  separate buffer %.4f ms/step   in place %.4f ms/step" % (rnd, run(other), run(frames)))
for rnd in range(2):
    print("round %d: DwtDctSvd         separate buffer %.4f ms/step   in place %.4f ms/step" % (rnd, run(other, svd=True), run(frames, svd=True)))
