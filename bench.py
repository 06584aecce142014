#!/usr/bin/env python3
"""bench.py -- 1080p frames/s for DCT watermark embed+detect on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames resident in HBM:
embed every frame, detect the produced frames, recover each frame's payload, all-gather the
payloads over the ranks (RCCL; a no-op on one GPU) and take the cross-frame vote.
Workload at N=1: BASELINE.json configs[1] -- 300 synthetic 1080p frames, payload
[0,1,1,0,0,1,0,1], Shuffler(key=0), alpha=20.  With N ranks every rank holds its own 300 frames
(weak scaling, frames shard with no data-path collective).

  python bench.py --gpus 1 --steps 100 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused mark+verify kernel:
it re-reads each frame, writes the marked frame and analyzes it).  Launch durations (`kernels`) come
from HIP event pairs the library attaches to every kernel dispatch of the timed steps
(hipExtLaunchKernelGGL start/stop events on the launch stream: the dispatch's own timestamps, no marker
packets, no measurable cost).  `cpu_baseline` is the plain-C restatement of the reference algorithm
(oracle/offmark_oracle.c, bit-identical to the NumPy oracle and the reference-run golden vectors;
OpenCV is not installed, so the reference itself cannot run) with one OpenMP thread per frame on the
host cores this process may use.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is achievable
PAYLOAD = np.array([0, 1, 1, 0, 0, 1, 0, 1])


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=300, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--chunk", type=int, default=0, help="frames per internal chunk (0 = engine default)")
    ap.add_argument("--alpha", type=float, default=20.0)
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2 = alternate steps between two HIP streams (own workspace and output buffer each)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="with one rank: still create the process group and issue every collective of the N>1 path "
                         "(1-rank RCCL all-gather on the side stream, barriers, MAX all-reduce) -- a dry run of that code")
    ap.add_argument("--separate-detect", action="store_true",
                    help="embed, then detect the written frames with the stand-alone detect kernels (analyze runs twice, "
                         "12 B/px of traffic) instead of the fused mark+verify kernel; same results bit for bit")
    ap.add_argument("--onepass", type=int, default=-1, metavar="GRID",
                    help="DCT codec: use the persistent one-pass embed+verify kernel with GRID workgroups (0 = its default)")
    ap.add_argument("--codec", choices=["dct", "dwtdctsvd"], default="dct",
                    help="dct = the BASELINE.json hot path (default); dwtdctsvd = the codec mark.py/detect.py construct")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (a one-GPU box cannot run RCCL across ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not attach HIP events to the kernel launches (roofline becomes null)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


def usable_cores():
    """Cores this process may actually use: the scheduler affinity, cut down to the cgroup CPU quota if one is set
    (a one-GPU box exposes all of the host's CPUs but grants a share of them)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = text[0], float(text[1])
            else:
                quota, period = text[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(-(-float(quota) // period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(cores, 64)


def cpu_baseline(frames_u8, wm, alpha, budget_s):
    """The oracle on the host cores, embed+detect on a bounded sample of the same workload.
    Primary figure: the C restatement (oracle/offmark_oracle.c, bit-identical to the NumPy oracle), one OpenMP
    thread per frame on every core this process may use.  The vectorised NumPy oracle on one core is timed on a
    few frames as well and quoted in `sample` (the reference itself is single-threaded Python + OpenCV)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import offmark_oracle as orc
    threads = usable_cores()
    n = len(frames_u8)
    done, ok, t0 = 0, True, time.perf_counter()
    while True:
        marked, used = c_oracle.mark_frames(frames_u8, wm, alpha=alpha, legacy=True, threads=threads)
        bits, _ = c_oracle.check_frames(marked, alpha=alpha, legacy=True, threads=threads)
        ok &= all(np.array_equal(orc.deshuffle(b, PAYLOAD.size, 0), PAYLOAD) for b in bits[:: max(1, n // 8)])
        done += n
        el = time.perf_counter() - t0
        if el + el * n / done > budget_s:
            break
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    enc = orc.DctEncoderOracle(alpha=alpha)
    enc.read_wm(wm)
    k = 3
    for i in range(k):
        orc.check_frame(orc.mark_frame(frames_u8[i], enc), orc.DctDecoderOracle(alpha=alpha))
    numpy_fps = k / (time.perf_counter() - t1)
    return dict(value=done / el, unit="frames/s", cores=used, kind="port",
                sample=f"{done} frame passes ({n} distinct {frames_u8.shape[2]}x{frames_u8.shape[1]} frames of the workload), "
                       f"embed+detect, C restatement of the reference algorithm with OpenMP over frames on {used} threads, "
                       f"{el:.1f} s; payload recovered: {ok}; vectorised NumPy oracle on 1 core: {numpy_fps:.1f} frames/s")


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from offmark import _hip
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.vote import gather_payloads, init_from_env, vote_segments
    from offmark.engine import DctEngine, default_chunk_frames
    from offmark.generator.shuffler import Shuffler
    from offmark.synthetic import synthetic_frames

    if a.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world = init_from_env(a.backend, force=a.rehearse_collectives)
    grouped = world > 1 or a.rehearse_collectives          # a process group exists: run the collectives
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _hip.load()
    if a.separate_detect:
        lib.ofmk_set_fused_verify(0)
    if a.onepass >= 0:
        lib.ofmk_set_fused_verify(3)
        lib.ofmk_set_onepass_grid(a.onepass)

    n, H, W = a.frames, a.height, a.width
    N = H * W // 64
    frames = synthetic_frames(n, H, W, seed=2000 + rank, device=dev)
    out = torch.empty_like(frames)
    wm = Shuffler(key=0).generate_wm(PAYLOAD, (1, N))
    wm_dev = torch.from_numpy(wm.astype(np.uint8)).to(dev)
    deg = DeShuffler(key=0).set_shape(PAYLOAD.shape)
    chunk = a.chunk or default_chunk_frames(H, W)
    eng = DctEngine(device=dev, chunk_frames=chunk)
    lanes = [dict(eng=eng, out=out, stream=torch.cuda.current_stream())]
    if a.streams == 2:
        lanes.append(dict(eng=DctEngine(device=dev, chunk_frames=chunk), out=torch.empty_like(frames), stream=torch.cuda.Stream()))
    seg_ids = np.repeat(np.arange(world), n)             # one segment per rank

    perm_dev = torch.as_tensor(deg.payload_idx, dtype=torch.int32).to(dev)
    host = [torch.empty((world * n, PAYLOAD.size), dtype=torch.uint8).pin_memory() for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]

    side = torch.cuda.Stream()                    # all-gather + download: off the compute stream, so a slow
    handoff = [torch.cuda.Event() for _ in range(2)]   # peer never stalls this rank's next step

    host_s = {"enqueue": 0.0, "vote": 0.0}     # host-side seconds spent issuing work / voting (not waiting)

    def barrier():
        if a.backend == "nccl":
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()

    def enqueue(k):
        """GPU half of step k: embed, detect the marked frames, per-frame payloads; then, on a side stream,
        the all-gather of the payloads and their download into pinned memory."""
        t_in = time.perf_counter()
        lane = lanes[k % len(lanes)]
        e = lane["eng"]
        with torch.cuda.stream(lane["stream"]):
            if a.codec == "dct":
                _, counts, _ = e.embed_detect(frames, wm_dev, L=PAYLOAD.size, alpha=a.alpha, out=lane["out"])
            else:
                _, counts, _ = e.svd_embed_detect(frames, wm_dev, L=PAYLOAD.size, scale=15, out=lane["out"])
            mine = e.payloads(counts, N, perm_dev)                       # [n, L] uint8, on device
            handoff[k & 1].record()
        with torch.cuda.stream(side):
            side.wait_event(handoff[k & 1])
            mine.record_stream(side)
            if a.backend == "gloo" and grouped:                        # rehearsal: gloo gathers host tensors
                everyone = gather_payloads(mine.cpu(), equal_shards=True, force=grouped)
            else:
                everyone = gather_payloads(mine, equal_shards=True, force=grouped)      # RCCL all-gather (N > 1)
            host[k & 1].copy_(everyone, non_blocking=True)
            ready[k & 1].record()
        host_s["enqueue"] += time.perf_counter() - t_in
        return mine

    def finish(k):
        """Host half of step k: the reference's cross-frame Counter vote, once its payloads have landed.
        It runs while the GPU is already working on step k+1 (double-buffered)."""
        ready[k & 1].synchronize()
        t_in = time.perf_counter()
        v = vote_segments(host[k & 1].numpy(), seg_ids)
        host_s["vote"] += time.perf_counter() - t_in
        return v

    def run(steps):
        votes, mine = None, None
        for k in range(steps):
            mine = enqueue(k)
            if k:
                votes = finish(k - 1)
        return finish(steps - 1), mine

    def fence():
        torch.cuda.synchronize()
        if grouped:
            barrier()
        torch.cuda.synchronize()

    # one-time setup, not a workload step: allocate the scratch for the chunk size in use and let the runtime load
    # the code objects (one full-size pass, so that profiles only ever see full-size launches); even --warmup 0 then
    # times steady-state steps
    for lane in lanes:
        lane["eng"].workspace(H, W, lane["eng"]._chunk(n, H, W))
        with torch.cuda.stream(lane["stream"]):
            if a.codec == "dct":
                _, c1, _ = lane["eng"].embed_detect(frames, wm_dev, L=PAYLOAD.size, alpha=a.alpha, out=lane["out"])
            else:
                _, c1, _ = lane["eng"].svd_embed_detect(frames, wm_dev, L=PAYLOAD.size, scale=15, out=lane["out"])
            p1 = lane["eng"].payloads(c1, N, perm_dev)
    torch.cuda.synchronize()
    if grouped:                                     # first collective on the side stream: RCCL sets its channels up here
        with torch.cuda.stream(side):
            gather_payloads(p1.cpu() if a.backend == "gloo" else p1, equal_shards=True, force=grouped)
        torch.cuda.synchronize()
    if a.warmup:
        run(a.warmup)
    n_chunks = (n + chunk - 1) // chunk
    launches_per_step = 5 * n_chunks                        # upper bound (3 with the fused verify kernel)
    use_events = not a.no_kernel_events
    KINDS = ("analyze", "finalize", "mark", "mark_fused", "svd")
    DOMINANT = ("mark" if a.separate_detect else "mark_fused") if a.codec == "dct" else "svd"

    def collect():
        ms = (ctypes.c_double * 5)()
        cnt = (ctypes.c_int * 5)()
        _hip.check(lib.ofmk_timing_collect(ms, cnt))
        lib.ofmk_timing_disable()
        return {k: dict(ms_total=ms[i], launches=cnt[i]) for i, k in enumerate(KINDS)}

    timed_steps = min(a.steps, 2000)                        # event pairs are pre-created; bound their number
    if use_events:                                          # every kernel of the timed steps carries its own event pair
        _hip.check(lib.ofmk_timing_enable(launches_per_step * timed_steps + 16, 0x1F))
    fence()
    host_s.update(enqueue=0.0, vote=0.0)
    t0 = time.perf_counter()
    votes, mine = run(a.steps)
    fence()
    elapsed = time.perf_counter() - t0
    host_ms = {k: round(1e3 * v / a.steps, 4) for k, v in host_s.items()}
    if grouped:
        t = torch.tensor([elapsed], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern = None
    if use_events:
        kern = collect()                       # per-launch durations from the timed region itself
        for v in kern.values():
            v["in_timed_region"] = True

    # correctness of what was timed: every frame's payload, every segment's vote
    payload_ok = bool((mine.cpu().numpy() == PAYLOAD[None]).all())
    votes_ok = all(v[0] is not None and np.array_equal(v[0], PAYLOAD) for v in votes.values())
    ber = float((mine.cpu().numpy() != PAYLOAD[None]).mean())

    if rank != 0:
        if world > 1:
            barrier()
            dist.destroy_process_group()
        return

    # achievable HBM bandwidth of this device, same run: 16-byte streaming copy, read + write
    nbytes = frames.numel() // 16 * 16
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = _hip.current_stream()
    for _ in range(2):
        _hip.check(lib.ofmk_hbm_copy(frames.data_ptr(), out.data_ptr(), nbytes, s))
    e0.record()
    for _ in range(5):
        _hip.check(lib.ofmk_hbm_copy(frames.data_ptr(), out.data_ptr(), nbytes, s))
    e1.record()
    torch.cuda.synchronize()
    copy_gbps = 5 * 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9

    fps = world * n * a.steps / elapsed
    frame_bytes = 3 * H * W
    roof = None
    extra = {}
    if kern:
        # algorithmic bytes per frame and kernel: analyze reads the frame (3 B/px);
        # mark reads it again and writes the marked frame (6 B/px); the fused mark+verify kernel
        # moves the same 6 B/px and spares detect's 3 B/px read.  Sum over a step = 9 B/px.
        alg = {"analyze": frame_bytes, "mark": 2 * frame_bytes, "mark_fused": 2 * frame_bytes, "svd": 2 * frame_bytes}
        names = {"analyze": "analyze_kernel<rgb8>", "mark": "mark_rgb8_kernel", "mark_fused": "mark_rgb8_kernel<fused verify>",
                 "svd": "svd_rgb8_kernel<embed+verify>"}
        per = {}
        for k, v in kern.items():
            if not v["launches"]:
                continue
            avg_ms = v["ms_total"] / v["launches"]
            passes = 2 if (k == "analyze" and not kern["mark_fused"]["launches"]) else 1
            d = dict(avg_launch_ms=round(avg_ms, 5), launches=v["launches"], ms_per_step=round(avg_ms * n_chunks * passes, 4),
                     timed_region=bool(v.get("in_timed_region", False)))
            if k in alg:
                frames_per_launch = n / n_chunks
                d["algorithmic_bytes_per_launch"] = int(frames_per_launch * alg[k])
                d["achieved_GBps"] = round(frames_per_launch * alg[k] / (avg_ms * 1e-3) / 1e9, 1)
            per[k] = d
        extra["kernels"] = per
        extra["kernel_ms_per_step"] = round(sum(v["ms_per_step"] for v in per.values()), 4)
        dom = DOMINANT if DOMINANT in per else max((k for k in per if k in alg), key=lambda k: per[k]["ms_per_step"])
        achieved = per[dom]["achieved_GBps"]
        traffic = None                          # PMC-measured HBM bytes per launch (separate rocprofv3 passes)
        tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if (tj["height"], tj["width"]) == (H, W) and dom in tj:
                per_frame = (tj[dom]["fetch_bytes"] + tj[dom]["write_bytes"]) / tj["frames_per_dispatch"]
                traffic = int(per_frame * per[dom]["algorithmic_bytes_per_launch"] / alg[dom])
        roof = dict(bound="hbm", kernel=names[dom], achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic,
                    algorithmic_bytes_per_launch=per[dom]["algorithmic_bytes_per_launch"],
                    avg_launch_ms=per[dom]["avg_launch_ms"], launches=per[dom]["launches"],
                    frac_of_measured_copy=round(achieved / copy_gbps, 4))

    base = None
    if world == 1 and not a.no_cpu_baseline and a.codec == "dct":
        try:
            base = cpu_baseline(frames[:96].cpu().numpy(), wm, a.alpha, a.cpu_seconds)
        except Exception as exc:                       # e.g. no C compiler on the box: report, do not lose the GPU line
            base = dict(value=None, unit="frames/s", cores=0, kind="port", sample=f"cpu baseline failed: {exc!r}")

    path_gbps = fps * 9 * H * W / 1e9                                   # 9 B/px per embed+detect frame
    line = {
        "metric": "1080p frames/sec embed+detect" if (H, W) == (1080, 1920) else f"{W}x{H} frames/sec embed+detect",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic {W}x{H} u8 RGB x{n} frames per GPU, "
                               f"{'DCT' if a.codec == 'dct' else 'DwtDctSvd'} embed+detect+vote "
                               f"(BASELINE.json configs[{2 if H >= 2160 else 1}])", "codec": a.codec,
                   "frames_per_gpu": n, "payload_bits": int(PAYLOAD.size), "alpha": a.alpha,
                   "chunk_frames": chunk,
                   "detect": ("separate kernels" if a.separate_detect else "one-pass kernel" if a.onepass >= 0
                              else "fused into the mark kernel") if a.codec == "dct" else "fused into the embed kernel",
                   "sharding": f"frames, {world} rank(s), one RCCL all-gather of payloads"},
        "payload_ber": ber, "payload_bit_exact": payload_ok and votes_ok,
        "roofline": roof,
        "path": {"algorithmic_GBps": round(path_gbps, 1), "bytes_per_frame": 9 * H * W,
                 "frac_of_peak": round(path_gbps / (HBM_PEAK_GBPS * world), 4),
                 "frac_of_measured_copy": round(path_gbps / (copy_gbps * world), 4)},
        "hbm_copy_GBps": round(copy_gbps, 1),
        "host_ms_per_step": host_ms,            # rank 0's CPU time issuing a step / voting on one; must stay < ms_per_step
        "cpu_baseline": base,
    }
    line.update(extra)
    print(json.dumps(line), flush=True)
    if grouped:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
