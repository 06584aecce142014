"""PCIe-inclusive throughput of embed(+verify): frames start and end in pinned HOST memory.
Three-stage pipeline on three streams (H2D, kernels, D2H), double-buffered device batches.
Not the benchmark (bench.py times HBM-resident inputs); this documents what the plugin boundary costs
when frames have to cross PCIe.  usage: python tools/pcie_pipeline.py [frames] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark.engine import DctEngine
from offmark.generator.shuffler import Shuffler
from offmark.synthetic import synthetic_frames

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 50
H, W = 1080, 1920
eng = DctEngine()
wm = torch.from_numpy(Shuffler(key=0).generate_wm(np.array([0, 1, 1, 0, 0, 1, 0, 1]), (1, H * W // 64)).astype(np.uint8)).cuda()
src = synthetic_frames(B, H, W, seed=1).cpu()
host_in = torch.empty((n, H, W, 3), dtype=torch.uint8).pin_memory()
for i in range(0, n, B):
    host_in[i:i + B] = src[: min(B, n - i)]
host_out = torch.empty_like(host_in).pin_memory()
dev_in = [torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
dev_out = [torch.empty_like(dev_in[0]) for _ in range(2)]
s_h2d, s_k, s_d2h = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev_in = [torch.cuda.Event() for _ in range(2)]
ev_k = [torch.cuda.Event() for _ in range(2)]
ev_out = [torch.cuda.Event() for _ in range(2)]

def run():
    for k, i in enumerate(range(0, n, B)):
        b, m = k & 1, min(B, n - i)
        with torch.cuda.stream(s_h2d):
            s_h2d.wait_event(ev_k[b])                     # the kernel that last read this buffer is done
            dev_in[b][:m].copy_(host_in[i:i + m], non_blocking=True)
            ev_in[b].record()
        with torch.cuda.stream(s_k):
            s_k.wait_event(ev_in[b]); s_k.wait_event(ev_out[b])
            eng.embed_detect(dev_in[b][:m], wm, L=8, out=dev_out[b][:m])
            ev_k[b].record()
        with torch.cuda.stream(s_d2h):
            s_d2h.wait_event(ev_k[b])
            host_out[i:i + m].copy_(dev_out[b][:m], non_blocking=True)
            ev_out[b].record()
    torch.cuda.synchronize()

run()
t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
gb = n * H * W * 3 / 1e9
print(f"PCIe-inclusive embed+verify: {n / dt:.0f} frames/s  ({gb / dt:.1f} GB/s each way, batch {B}, {n} frames of {W}x{H})")
