"""DwtDctSvd embed + verify + payloads: plain counts (zero-fill dispatch + global atomics) against partial counts
(OFMK_F_PARTIAL_COUNTS: per-workgroup sums stored, added up inside the payload kernel) -- interleaved in one process.
usage (GPU box): python tools/svd_partial_ab.py [frames=300] [rounds=6] [reps=20]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from offmark.degenerator.de_shuffler import DeShuffler  # noqa: E402
from offmark.engine import DctEngine  # noqa: E402
from offmark.generator.shuffler import Shuffler  # noqa: E402
from offmark.synthetic import synthetic_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H, W, L = 1080, 1920, 8
P = np.array([0, 1, 1, 0, 0, 1, 0, 1])
dev = torch.device("cuda", 0)
frames = synthetic_frames(n, H, W, seed=2000, device=dev)
out = torch.empty_like(frames)
wm = torch.from_numpy(Shuffler(key=0).generate_wm(P, (1, H * W // 64)).astype(np.uint8)).to(dev)
perm = torch.as_tensor(DeShuffler(key=0).set_shape(P.shape).payload_idx, dtype=torch.int32).to(dev)
e = DctEngine(device=dev)
res = {}
for blk in (4, 8):
    nb = DctEngine.svd_bits_per_frame(H, W, blk)
    tiles = e.lib.ofmk_svd_count_tiles(H, W, blk)
    bufs = {False: torch.empty((n, L), dtype=torch.int32, device=dev), True: torch.empty((n, tiles, L), dtype=torch.int32, device=dev)}
    pay = {False: torch.empty((n, L), dtype=torch.uint8, device=dev), True: torch.empty((n, L), dtype=torch.uint8, device=dev)}

    def step(partial):
        _, c, _ = e.svd_embed_detect(frames, wm, L, out=out, blk=blk, counts=bufs[partial], partial=partial)
        e.payloads(c, nb, perm, out=pay[partial])
    for p in (False, True):
        step(p)
    torch.cuda.synchronize()
    assert torch.equal(pay[False], pay[True]) and (pay[True].cpu().numpy() == P[None]).all()
    for _ in range(200 if blk == 4 else 0):
        step(True)                                            # device out of idle
    t = {False: [], True: []}
    for r in range(rounds):
        for p in (False, True, True, False):
            step(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                step(p)
            torch.cuda.synchronize()
            t[p].append((time.perf_counter() - t0) / reps)
    res[blk] = {p: float(np.median(v)) for p, v in t.items()}
    a, b = res[blk][False], res[blk][True]
    print(f"blk {blk}: plain counts {1e3 * a:.4f} ms/step ({n / a / 1e3:.1f} k frames/s)   partial counts {1e3 * b:.4f} ms/step ({n / b / 1e3:.1f} k frames/s)   "
          f"{100 * (a / b - 1):+.2f} %")
