"""Launch durations of the DwtDctSvd kernels, blk = 4 against blk = 8, detect / embed / embed+verify (300 x 1080p)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark.engine import DctEngine
from offmark.generator.shuffler import Shuffler
from offmark.synthetic import synthetic_frames
n, H, W = 300, 1080, 1920
eng = DctEngine()
frames = synthetic_frames(n, H, W, seed=2000)
out = torch.empty_like(frames)
wm = torch.from_numpy(Shuffler(key=0).generate_wm(np.array([0, 1, 1, 0, 0, 1, 0, 1]), (1, H * W // 64)).astype(np.uint8)).cuda()
def t(fn, k=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for blk in (4, 8):
    for sc in (None, [9, 15, 21]):
        d = t(lambda: eng.svd_detect(frames, 8, blk=blk, scales=sc))
        e = t(lambda: eng.svd_embed(frames, wm, out=out, blk=blk, scales=sc))
        v = t(lambda: eng.svd_embed_detect(frames, wm, 8, out=out, blk=blk, scales=sc))
        print(f"blk={blk} scales={sc}: detect {d:.3f} ms  embed {e:.3f} ms  embed+verify {v:.3f} ms  (300 x 1080p)")
