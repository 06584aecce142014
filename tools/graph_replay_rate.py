"""The embed+detect+payloads step of bench.py config 2 as a captured hipGraph against plain stream launches (300 x 1080p)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark.degenerator.de_shuffler import DeShuffler
from offmark.engine import DctEngine
from offmark.generator.shuffler import Shuffler
from offmark.synthetic import synthetic_frames
n, H, W = 300, 1080, 1920
N = H * W // 64
P = np.array([0, 1, 1, 0, 0, 1, 0, 1])
eng = DctEngine()
frames = synthetic_frames(n, H, W, seed=2000)
out = torch.empty_like(frames)
wm = torch.from_numpy(Shuffler(key=0).generate_wm(P, (1, N)).astype(np.uint8)).cuda()
perm = torch.as_tensor(DeShuffler(key=0).set_shape(P.shape).payload_idx, dtype=torch.int32).cuda()
counts = torch.empty((n, 8), dtype=torch.int32, device="cuda")
payload = torch.empty((n, 8), dtype=torch.uint8, device="cuda")
eng.workspace(H, W, n)
def step():
    _, c, _ = eng.embed_detect(frames, wm, L=8, out=out)
    return eng.payloads(c, N, perm, out=payload)
def rate(fn, k=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return n * k / (time.perf_counter() - t0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    for rep in range(3):
        a = rate(step); b = rate(g.replay)
        print(f"stream launches {a:9.0f} frames/s   graph replay {b:9.0f} frames/s   ({100 * (b / a - 1):+.1f} %)  payload ok {bool((payload.cpu().numpy() == P).all())}")
