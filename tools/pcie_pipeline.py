"""PCIe-inclusive throughput of embed(+verify): frames start and end in pinned HOST memory.
Three-stage pipeline on three streams (H2D, kernels, D2H), double-buffered device batches.
Not the benchmark's `value` (bench.py times HBM-resident inputs); this documents what the boundary costs when
frames have to cross PCIe, for the two frame formats the engine takes:
  rgb24  interleaved u8 RGB, 3 B/px each way (what the reference's ffmpeg pipes carry, frame_reader.py:42-64)
  i420   planar 4:2:0, 1.5 B/px each way (what a decoder produces / an encoder takes, frame_writer.py:33-34)
bench.py calls measure() and reports both in its line's `pcie_inclusive` extras.
usage: python tools/pcie_pipeline.py [frames] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))


def measure(fmt="rgb24", n=300, B=50, H=1080, W=1920, eng=None):
    """frames/s of embed+verify with every frame crossing PCIe in and out.  Returns (fps, GB/s each way)."""
    import numpy as np
    import torch
    from offmark.engine import DctEngine
    from offmark.generator.shuffler import Shuffler
    from offmark.synthetic import synthetic_frames
    eng = eng or DctEngine()
    wm = torch.from_numpy(Shuffler(key=0).generate_wm(np.array([0, 1, 1, 0, 0, 1, 0, 1]), (1, H * W // 64)).astype(np.uint8)).cuda()
    src = synthetic_frames(B, H, W, seed=1)
    if fmt == "i420":
        src = eng.rgb_to_yuv420(src)
    src = src.cpu()
    shape = tuple(src.shape[1:])
    host_in = torch.empty((n,) + shape, dtype=torch.uint8).pin_memory()
    for i in range(0, n, B):
        host_in[i:i + B] = src[: min(B, n - i)]
    host_out = torch.empty_like(host_in).pin_memory()
    dev_in = [torch.empty((B,) + shape, dtype=torch.uint8, device="cuda") for _ in range(2)]
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
                s_k.wait_event(ev_in[b])
                s_k.wait_event(ev_out[b])
                if fmt == "i420":
                    eng.embed_detect_yuv420(dev_in[b][:m], H, W, wm, 8, out=dev_out[b][:m])
                else:
                    eng.embed_detect(dev_in[b][:m], wm, L=8, out=dev_out[b][:m])
                ev_k[b].record()
            with torch.cuda.stream(s_d2h):
                s_d2h.wait_event(ev_k[b])
                host_out[i:i + m].copy_(dev_out[b][:m], non_blocking=True)
                ev_out[b].record()
        torch.cuda.synchronize()

    run()
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    return n / dt, n * host_in[0].numel() / 1e9 / dt


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    for fmt in ("rgb24", "i420"):
        fps, gbps = measure(fmt, n, B)
        print(f"PCIe-inclusive embed+verify, {fmt}: {fps:.0f} frames/s  ({gbps:.1f} GB/s each way, batch {B}, {n} frames of 1920x1080)")
