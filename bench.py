#!/usr/bin/env python3
"""bench.py -- 1080p frames/s for DCT watermark embed+detect on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic frames resident in HBM:
embed every frame, detect the produced frames, recover each frame's payload, all-gather the
payloads over the ranks (RCCL; a no-op on one GPU) and take the cross-frame vote.

  --config 2 (default)  BASELINE.json configs[1]: 300 synthetic 1080p frames per GPU, payload
                        [0,1,1,0,0,1,0,1], Shuffler(key=0), alpha=20.  Weak scaling: every rank holds its
                        own 300 frames (frames shard with no data-path collective).
  --config 3            configs[2]: 1000 synthetic 4K frames per GPU, processed in internal chunks.
  --config 4            configs[3]: 8 segments x 48 frames of 1080p, segment s carries format(s % 256, '08b')
                        (tests/segment_mark_detect_hls.py:42-55), segments sharded over the ranks (strong
                        scaling), per-segment Counter vote on every rank.
  --config 5            configs[4]: leak identification (tests/generate_leak.py:59-108 +
                        tests/detect_watermarks.py:321-364): 8 segments x 3 copies are marked in set-up
                        (payload = segment(4b)||copy(4b)), a leak takes one copy per segment plus the
                        build-defined re-quantisation attack; the timed step is the batched DETECT of the
                        leak's frames, the all-gather of the payloads and the vote -> copy sequence.

  python bench.py --gpus 1 --steps 100 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...      with N > 1 and no launcher (WORLD_SIZE unset): bench.py starts that same
                                    torch.distributed.run command itself as a child process, before anything
                                    touches the GPU, and relays rank 0's line and the exit code.
`collective.ranks` / `rccl_ranks` = what an all-reduce of ones returned: the ranks the collective library really joined.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (config 2: the fused mark+verify kernel:
it re-reads each frame, writes the marked frame and analyzes it).  Launch durations (`kernels`) come
from HIP event pairs the library attaches to every kernel dispatch of the timed steps
(hipExtLaunchKernelGGL start/stop events on the launch stream: the dispatch's own timestamps, no marker
packets, no measurable cost).  `roofline.traffic` (PMC-measured HBM bytes per launch) is taken from
profiles/ only when that profile was made from exactly the kernel sources that are running (hash stamp),
else null.  `cpu_baseline` is the plain-C restatement of the reference algorithm (oracle/offmark_oracle.c,
bit-identical to the NumPy oracle and the golden vectors; kind "port": OpenCV is not installed, so the
reference itself cannot run, and OpenCV's own float rounding is parity-unpinned) with one OpenMP thread per
frame on the host cores this process may use; `cpu_baseline.variants` adds BASELINE.md's A / B1 / B2 forms.
"""
import argparse
import hashlib
import json
import os
import subprocess
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
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json workload (see module text)")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (configs 4/5: per segment); 0 = the config's own")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
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
    ap.add_argument("--codec", choices=["dct", "dwtdctsvd"], default="dct",
                    help="dct = the BASELINE.json hot path (default); dwtdctsvd = the codec mark.py/detect.py construct")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (a one-GPU box cannot run RCCL across ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not attach HIP events to the kernel launches (roofline becomes null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the separate-detect side measurement")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget of the CPU baseline's main (C, all cores) sample")
    return ap.parse_args()


def source_sha16():
    """Hash of everything the library is compiled from: stamps profiles so a stale one is never quoted."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "video-fingerprinting_amd", "csrc")
    for path in sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, "include", "offmark_hip.h")]:
        h.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]


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
    value: the C restatement (oracle/offmark_oracle.c, bit-identical to the NumPy oracle), one OpenMP thread per
    frame on every core this process may use.  variants: BASELINE.md section 3's forms of the NumPy oracle --
    A reference-shaped per-block Python loop on one core, B1 all-blocks-at-once NumPy on one core, B2 = B1 in one
    worker process per core (oracle/cpu_baseline_worker.py; separate processes that never touch the GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import offmark_oracle as orc
    threads = usable_cores()
    n = len(frames_u8)
    H, W = frames_u8.shape[1:3]
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
    variants = {}
    try:
        # B1: vectorised NumPy, one core
        enc = orc.DctEncoderOracle(alpha=alpha)
        enc.read_wm(wm)
        t1, k = time.perf_counter(), 3
        for i in range(k):
            orc.check_frame(orc.mark_frame(frames_u8[i], enc), orc.DctDecoderOracle(alpha=alpha))
        variants["B1_numpy_vectorised"] = dict(value=round(k / (time.perf_counter() - t1), 2), unit="frames/s", cores=1, frames=k)
        # A: reference-shaped per-block Python loop, one core, on a quarter-frame crop scaled by its block count
        crop = np.ascontiguousarray(frames_u8[0][: H // 16 * 8, : W // 16 * 8])
        ch, cw = crop.shape[:2]
        encl = orc.DctEncoderOracle(alpha=alpha, form="loop")
        encl.read_wm(orc.shuffle_generate(PAYLOAD, (1, ch * cw // 64), 0))
        t2 = time.perf_counter()
        orc.check_frame(orc.mark_frame(crop, encl), orc.DctDecoderOracle(alpha=alpha, form="loop"))
        ta = (time.perf_counter() - t2) * ((H // 8) * (W // 8)) / ((ch // 8) * (cw // 8))
        variants["A_reference_shaped_loop"] = dict(value=round(1.0 / ta, 3), unit="frames/s", cores=1,
                                                   frames=round((ch * cw) / (H * W), 3),
                                                   note=f"timed on a {cw}x{ch} crop, scaled by the block count")
    except Exception as exc:
        variants["error_A_B1"] = repr(exc)
    # B2: B1 fanned out, one worker process per core
    try:
        per = 2
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline_worker.py"), str(H), str(W),
                                   str(per), str(3000 + i), str(alpha)], stdout=subprocess.PIPE, text=True, env=env)
                 for i in range(threads)]
        spans = [json.loads(p.communicate(timeout=300)[0].strip().splitlines()[-1]) for p in procs]
        wall = max(s["t1"] for s in spans) - min(s["t0"] for s in spans)
        variants["B2_numpy_vectorised_all_cores"] = dict(value=round(threads * per / wall, 2), unit="frames/s", cores=threads,
                                                         frames=threads * per, payload_ok=all(s["ok"] for s in spans))
    except Exception as exc:                                   # report, never lose the line
        variants["B2_numpy_vectorised_all_cores"] = dict(value=None, error=repr(exc))
    return dict(value=done / el, unit="frames/s", cores=used, kind="port",
                sample=f"{done} frame passes ({n} distinct {W}x{H} frames of the workload), embed+detect, C restatement of the "
                       f"reference algorithm with OpenMP over frames on {used} threads, {el:.1f} s; payload recovered: {ok}. "
                       "The reference itself needs OpenCV (absent): DCT/colour primitives are restated, OpenCV's float "
                       "rounding is parity-unpinned",
                variants=variants)


def attack_suite(torch, eng, clean, per_seg, payloads, chosen, fp, vote_segments, deg, N, alpha, H, W):
    """BASELINE.json configs[4]: the leak's frames under the build-defined attacks of SURVEY 8d (none exist upstream: the
    reference's only lossy leg is a JPEG, tests/test.py:99, and its HLS re-encode).  `clean`: marked frames [S * per_seg, H, W, 3]
    on the device.  Per attack: payload bit error rate over the frames, frames decoded exactly, segments whose Counter vote is
    right, and whether the leak's copy sequence comes out.  Reported as measured: cropping moves the 8x8 grid and is EXPECTED to
    defeat a block-DCT QIM scheme (so is a strong re-quantisation such as JPEG quality 75); a mild rescale may or may not survive
    depending on the content; only "none" and "noise" are parity-gated (tests)."""
    import io
    S = len(payloads)
    seg = np.repeat(np.arange(S), per_seg)
    want = np.stack([payloads[s] for s in seg])

    def nchw(x):
        return x.permute(0, 3, 1, 2).float()

    def back(x):
        return x.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()

    def resize(x, h, w):
        return torch.nn.functional.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)

    def jpeg(x, q):
        from PIL import Image
        out = []
        for f in x.cpu().numpy():
            buf = io.BytesIO()
            Image.fromarray(f).save(buf, format="JPEG", quality=q, subsampling=2)
            out.append(np.array(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
        return torch.from_numpy(np.stack(out)).to(x.device)

    g = torch.Generator(device=clean.device).manual_seed(11)
    attacks = {
        "none": lambda x: x,
        "noise_sigma2": lambda x: (x.float() + 2.0 * torch.randn(x.shape, device=x.device, generator=g)).round().clamp(0, 255).to(torch.uint8),
        "scale_2_3_and_back": lambda x: back(resize(resize(nchw(x), H * 2 // 3, W * 2 // 3), H, W)),
        "crop16_and_resize_back": lambda x: back(resize(nchw(x)[:, :, 16:H - 16, 16:W - 16], H, W)),
        "jpeg_q95_420": lambda x: jpeg(x, 95),
        "jpeg_q75_420": lambda x: jpeg(x, 75),
    }
    out = {}
    for name, fn in attacks.items():
        try:
            counts, _ = eng.detect(fn(clean), PAYLOAD.size, alpha=alpha)
            got = deg.degenerate_counts(counts.cpu().numpy(), N)
            votes = vote_segments(got, seg)
            seg_ok = sum(int(v[0] is not None and np.array_equal(v[0], payloads[s])) for s, v in votes.items())
            out[name] = dict(payload_ber=round(float((got != want).mean()), 4), frames_exact=round(float((got == want).all(axis=1).mean()), 4),
                             segments_ok=f"{seg_ok}/{S}", copies_recovered=fp.identify_copies({s + 1: v for s, v in votes.items()}) == chosen)
        except Exception as exc:                                   # never lose the line over a side report
            out[name] = dict(error=repr(exc))
    out["note"] = (f"{per_seg} frames per segment, DCT codec; attacks are build-defined tensor ops between embed and detect (Pillow for the JPEGs); "
                   "a crop moves the 8x8 grid and is expected to fail for this scheme; the others are reported as measured")
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD
    `python -m torch.distributed.run` (one process per GPU over RCCL), relay rank 0's JSON line and the exit code.
    Nothing in this parent has touched the GPU (no torch import, no HIP call), and the parent is never replaced
    (no exec): the ranks are ordinary child processes.  Sharding shape: tests/segment_mark_detect_hls.py:407-412
    (independent segments), here one rank per GPU."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["OFMK_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for text in child.stdout:                                # rank 0 prints the one JSON line; anything else goes to stderr
        if text.lstrip().startswith("{"):
            lines.append(text)
        else:
            sys.stderr.write(text)
    rc = child.wait()
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    elif rc == 0:
        rc = 1                                               # a run without a line is a failed run
    raise SystemExit(rc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    import torch
    import torch.distributed as dist
    from offmark import _hip
    from offmark import fingerprint as fp
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.vote import gather_payloads, init_from_env, shard_range, vote_segments
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
    flags = _hip.F_SEPARATE_DETECT if a.separate_detect else 0
    ranks_seen = 1
    if grouped:                                   # how many ranks the collective library really connected: sum of ones
        one = torch.ones(1, dtype=torch.int32, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    # ---- workload ------------------------------------------------------------------------------------
    cfg = a.config
    H = a.height or (2160 if cfg == 3 else 1080)
    W = a.width or (3840 if cfg == 3 else 1920)
    N = H * W // 64
    deg = DeShuffler(key=0).set_shape(PAYLOAD.shape)
    S, F, C = 8, 48, 3                                       # configs 4/5: segments, frames per segment, copies
    chosen = None
    if cfg in (2, 3):
        scaling, equal = "weak", True
        n = a.frames or (1000 if cfg == 3 else 300)
        frames = synthetic_frames(n, H, W, seed=2000 + rank, device=dev)
        wm_table = Shuffler(key=0).generate_wm(PAYLOAD, (1, N)).astype(np.uint8)
        rows_local = None
        seg_global = np.repeat(np.arange(world), n)          # one "segment" per rank
        first = rank * n
        expected = {r: PAYLOAD for r in range(world)}
        total_frames = world * n
        mode = "embed_detect"
    else:
        scaling = "strong"                                   # the 8-segment job is split over the ranks
        if a.frames:
            F = a.frames
        s0, s1 = shard_range(S, rank, world)
        n, first = (s1 - s0) * F, s0 * F
        equal = S % world == 0
        seg_global = np.repeat(np.arange(S), F)
        total_frames = S * F
        src = synthetic_frames(max(n, 1), H, W, seed=4000 + rank, device=dev)[:n]
        if cfg == 4:
            # segment s carries format(s % 256, '08b'); numbering starts at 1 because segment 0's all-zero payload
            # cannot be decoded by the reference's mid-range threshold (de_shuffler.py:20-21)
            payloads = np.stack([fp.payload_for_segment(s + 1) for s in range(S)])
            wm_table = np.stack([Shuffler(key=0).generate_wm(p, (1, N))[0] for p in payloads]).astype(np.uint8)
            rows_local = np.repeat(np.arange(s0, s1), F).astype(np.int32)
            expected = {s: payloads[s] for s in range(S)}
            frames = src
            mode = "embed_detect"
        else:
            table = np.stack([Shuffler(key=0).generate_wm(fp.payload_for_segment(s + 1, c), (1, N))[0]
                              for s in range(S) for c in range(C)]).astype(np.uint8)
            chosen = fp.select_copies("01201201", S, C)
            rows = np.array([(s * C + chosen[s]) for s in range(s0, s1) for _ in range(F)], dtype=np.int32)
            setup = DctEngine(device=dev)
            leak = setup.embed(src, table, alpha=a.alpha, wm_row=rows) if n else src
            g = torch.Generator(device=dev).manual_seed(7 + rank)    # build-defined attack (i): N(0, 2) + round/clip
            frames = (leak.float() + 2.0 * torch.randn(leak.shape, device=dev, generator=g)).round().clamp(0, 255).to(torch.uint8)
            # a few clean frames of every local segment for the attack suite reported next to the line (not timed)
            keep = min(F, 8)
            leak_sample = leak.view(s1 - s0, F, H, W, 3)[:, :keep].reshape(-1, H, W, 3).clone() if n else None
            del leak, setup
            wm_table, rows_local = None, None
            expected = {s: fp.payload_for_segment(s + 1, chosen[s]) for s in range(S)}
            mode = "detect"
    out = torch.empty_like(frames) if mode == "embed_detect" else None
    wm_dev = torch.from_numpy(wm_table).to(dev) if wm_table is not None else None
    rows_dev = torch.from_numpy(rows_local).to(dev) if rows_local is not None else None
    chunk = a.chunk or default_chunk_frames(H, W)
    n_chunks = max(1, (n + chunk - 1) // chunk)
    timed_steps = min(a.steps, 2000)                        # event pairs are pre-created; bound their number
    DOMINANT = ("analyze" if mode == "detect" else "mark" if a.separate_detect else "mark_fused") if a.codec == "dct" else "svd"
    # event pairs on the DOMINANT kernel's launches only while `value` is timed (a pair on every launch of a step costs ~1.3 %
    # of the step: measured 267.5 k against 271.0 k frames/s, interleaved); the other kernels' durations come from a short pass
    # of their own after the timed region (`kernels`)
    timing = None if a.no_kernel_events else _hip.Timing(2 * n_chunks * timed_steps + 16, 1 << _hip.TIMING_KINDS.index(DOMINANT))
    opts_plain = _hip.Opts(flags, 0, None)
    opts_timed = timing.opts(flags) if timing else opts_plain
    lanes = [dict(eng=DctEngine(device=dev, chunk_frames=chunk, opts=opts_plain), out=out, stream=torch.cuda.current_stream())]
    if a.streams == 2:
        lanes.append(dict(eng=DctEngine(device=dev, chunk_frames=chunk, opts=opts_plain),
                          out=torch.empty_like(frames) if out is not None else None, stream=torch.cuda.Stream()))

    perm_dev = torch.as_tensor(deg.payload_idx, dtype=torch.int32).to(dev)
    host = [torch.empty((total_frames, PAYLOAD.size), dtype=torch.uint8).pin_memory() for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]

    side = torch.cuda.Stream()                    # all-gather + download: off the compute stream, so a slow
    handoff = [torch.cuda.Event() for _ in range(2)]   # peer never stalls this rank's next step

    host_s = {"enqueue": 0.0, "vote": 0.0}     # host-side seconds spent issuing work / voting (not waiting)

    def barrier():
        if a.backend == "nccl":
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()

    def hot_path(e, lane_out):
        """embed + detect (config 5: detect only) + per-frame payloads for this rank's frames."""
        if n == 0:
            return torch.empty((0, PAYLOAD.size), dtype=torch.uint8, device=dev)
        if mode == "detect":
            counts, _ = e.detect(frames, PAYLOAD.size, alpha=a.alpha)
        elif a.codec == "dct":
            _, counts, _ = e.embed_detect(frames, wm_dev, L=PAYLOAD.size, alpha=a.alpha, wm_row=rows_dev, out=lane_out)
        else:
            _, counts, _ = e.svd_embed_detect(frames, wm_dev, L=PAYLOAD.size, scale=15, wm_row=rows_dev, out=lane_out)
        return e.payloads(counts, N, perm_dev)                             # [n, L] uint8, on device

    def enqueue(k):
        """GPU half of step k; then, on a side stream, the all-gather of the payloads and their download into
        pinned memory."""
        t_in = time.perf_counter()
        lane = lanes[k % len(lanes)]
        with torch.cuda.stream(lane["stream"]):
            mine = hot_path(lane["eng"], lane["out"])
            handoff[k & 1].record()
        with torch.cuda.stream(side):
            side.wait_event(handoff[k & 1])
            mine.record_stream(side)
            if a.backend == "gloo" and grouped:                        # rehearsal: gloo gathers host tensors
                everyone = gather_payloads(mine.cpu(), equal_shards=equal, force=grouped)
            else:
                everyone = gather_payloads(mine, equal_shards=equal, force=grouped)      # RCCL all-gather (N > 1)
            host[k & 1].copy_(everyone, non_blocking=True)
            ready[k & 1].record()
        host_s["enqueue"] += time.perf_counter() - t_in
        return mine

    def finish(k):
        """Host half of step k: the reference's cross-frame Counter vote, once its payloads have landed.
        It runs while the GPU is already working on step k+1 (double-buffered)."""
        ready[k & 1].synchronize()
        t_in = time.perf_counter()
        v = vote_segments(host[k & 1].numpy(), seg_global)
        host_s["vote"] += time.perf_counter() - t_in
        return v

    def run(steps):
        mine = None
        for k in range(steps):
            mine = enqueue(k)
            if k:
                finish(k - 1)
        return finish(steps - 1), mine

    def fence():
        torch.cuda.synchronize()
        if grouped:
            barrier()
        torch.cuda.synchronize()

    def timed(steps):
        fence()
        host_s.update(enqueue=0.0, vote=0.0)
        t0 = time.perf_counter()
        votes, mine = run(steps)
        fence()
        el = time.perf_counter() - t0
        if grouped:
            t = torch.tensor([el], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, votes, mine

    # one-time setup, not a workload step: allocate the scratch for the chunk size in use and let the runtime load
    # the code objects (one full-size pass, so that profiles only ever see full-size launches); even --warmup 0 then
    # times steady-state steps
    for lane in lanes:
        if n:
            lane["eng"].workspace(H, W, lane["eng"]._chunk(n, H, W))
        with torch.cuda.stream(lane["stream"]):
            p1 = hot_path(lane["eng"], lane["out"])
    torch.cuda.synchronize()
    if grouped:                                     # first collective on the side stream: RCCL sets its channels up here
        with torch.cuda.stream(side):
            gather_payloads(p1.cpu() if a.backend == "gloo" else p1, equal_shards=equal, force=grouped)
        torch.cuda.synchronize()
    for lane in lanes:
        lane["eng"].opts = opts_timed                       # every kernel of the timed steps carries its own event pair ...
    if a.warmup:
        run(a.warmup)                                       # ... and so do the warm-up steps: they are the timed steps' twins
        if timing:
            torch.cuda.synchronize()
            timing.collect()                                # rewind the event pool: the durations reported are the timed region's
    elapsed, votes, mine = timed(a.steps)
    host_ms = {k: round(1e3 * v / a.steps, 4) for k, v in host_s.items()}
    for lane in lanes:
        lane["eng"].opts = opts_plain
    kern = None
    if timing:
        kern = timing.collect()                # the dominant kernel's per-launch durations from the timed region itself
        timing.close()
        if True:                               # every kernel kind, from a short pass of its own straight after
            kb = max(3, min(a.steps, 20))
            t_all = _hip.Timing(6 * n_chunks * kb + 16)
            o_all = t_all.opts(flags)
            for lane in lanes:
                lane["eng"].opts = o_all
            run(kb)
            torch.cuda.synchronize()
            for k_, v_ in t_all.collect().items():
                if k_ != DOMINANT or not kern[k_]["launches"]:
                    kern[k_] = v_
            for lane in lanes:
                lane["eng"].opts = opts_plain
            t_all.close()

    # correctness of what was timed: every frame's payload, every segment's vote (and the leak's copy sequence)
    want_mine = np.stack([expected[s] for s in seg_global[first:first + n]]) if n else np.zeros((0, PAYLOAD.size), np.int64)
    got_mine = mine.cpu().numpy()
    ber = float((got_mine != want_mine).mean()) if n else 0.0
    votes_ok = len(votes) == len(expected) and all(v[0] is not None and np.array_equal(v[0], expected[s]) for s, v in votes.items())
    payload_ok = bool((got_mine == want_mine).all())
    if cfg == 5:     # under the noise attack single frames may misread; what must hold is the vote -> copy sequence
        votes_ok = votes_ok and fp.identify_copies({s + 1: v for s, v in votes.items()}) == chosen
        payload_ok = votes_ok

    extra = {}
    if cfg == 5 and world == 1 and not a.no_extras and n:
        extra["attacks"] = attack_suite(torch, lanes[0]["eng"], leak_sample, keep, [expected[s] for s in range(S)], chosen, fp,
                                        vote_segments, deg, N, a.alpha, H, W)
    # the same K steps once more, straight after the timed region.  `value` is the contract's figure (W warm-up steps after
    # idle, then K steps: with a short K that sits on the device's clock ramp); this one is the rate the device settles at.
    if not a.no_extras:
        el_b, v_b, _ = timed(a.steps)
        extra["value_second_pass"] = round(total_frames * a.steps / el_b, 1)
        extra["second_pass"] = dict(steps=a.steps, ms_per_step=round(1e3 * el_b / a.steps, 4),
                                    votes_ok=len(v_b) == len(expected) and all(v[0] is not None and np.array_equal(v[0], expected[s])
                                                                               for s, v in v_b.items()),
                                    note="the same K steps again straight after the timed region (no event pairs on the launches)")

    # the same steps alternating between TWO HIP streams (own workspace and output buffer each): independent batches overlap, one
    # step's analyze beside the other's mark+verify, launch gaps and kernel tails filled.  Measured: +5 % over a single stream whose
    # every launch carries an event pair, 0 to +2 % over the eventless single stream (`value_second_pass`): most of what it hides is
    # instrumentation.  Reported next to `value`, which stays single-stream: under concurrency a kernel's launch duration includes
    # the time it shares the device, so the roofline object (bytes per launch / launch duration) would no longer describe the kernel.
    if cfg in (2, 3) and a.codec == "dct" and a.streams == 1 and not a.no_extras and n:
        try:
            lanes.append(dict(eng=DctEngine(device=dev, chunk_frames=chunk, opts=opts_plain),
                              out=torch.empty_like(frames) if out is not None else None, stream=torch.cuda.Stream()))
            lanes[1]["eng"].opts = lanes[0]["eng"].opts
            lanes[1]["eng"].workspace(H, W, lanes[1]["eng"]._chunk(n, H, W))
            run(2)
            el_t, v_t, _ = timed(a.steps)
            extra["value_two_streams"] = round(total_frames * a.steps / el_t, 1)
            extra["two_streams"] = dict(steps=a.steps, ms_per_step=round(1e3 * el_t / a.steps, 4),
                                        path_frac_of_peak=round(total_frames * a.steps / el_t * 9 * H * W / 1e9 / (HBM_PEAK_GBPS * world), 4),
                                        votes_ok=len(v_t) == len(expected) and all(v[0] is not None and np.array_equal(v[0], expected[s])
                                                                                   for s, v in v_t.items()),
                                        note="the same K steps, consecutive steps on two HIP streams (python bench.py --streams 2 times this form)")
        except Exception as exc:                               # e.g. no room for the second output buffer
            extra["two_streams"] = dict(error=repr(exc))
        finally:
            if len(lanes) > 1:
                torch.cuda.synchronize()
                lanes.pop()

    # second figure of the same line: SURVEY 8d config 2 read literally (embed, then the stand-alone detect)
    if cfg == 2 and a.codec == "dct" and not a.separate_detect and not a.no_extras:
        sep = _hip.Opts(_hip.F_SEPARATE_DETECT, 0, None)
        for lane in lanes:
            lane["eng"].opts = sep
        k2 = max(3, min(a.steps, 20))
        run(1)
        el2, v2, _ = timed(k2)
        for lane in lanes:
            lane["eng"].opts = opts_plain
        extra["value_separate_detect"] = round(world * n * k2 / el2, 1)
        extra["separate_detect"] = dict(steps=k2, ms_per_step=round(1e3 * el2 / k2, 4),
                                        note="embed, then the stand-alone detect on the written frames (12 B/px real traffic); "
                                             "bit-identical results",
                                        votes_ok=all(np.array_equal(v[0], expected[s]) for s, v in v2.items()))

    # planar 4:2:0 frames through the same step, HBM-resident: what the fused ingest/egress costs or saves
    if cfg == 2 and a.codec == "dct" and not a.separate_detect and not a.no_extras and H % 8 == 0 and W % 8 == 0:
        e0 = lanes[0]["eng"]
        planes = e0.rgb_to_yuv420(frames)
        pout = torch.empty_like(planes)

        def planar_step():
            _, c, _ = e0.embed_detect_yuv420(planes, H, W, wm_dev, PAYLOAD.size, alpha=a.alpha, out=pout)
            return e0.payloads(c, N, perm_dev)
        planar_step()
        fence()
        t0 = time.perf_counter()
        k3 = max(3, min(a.steps, 20))
        for _ in range(k3):
            pm = planar_step()
        fence()
        el3 = time.perf_counter() - t0
        extra["planar_i420"] = dict(value=round(world * n * k3 / el3, 1), unit="frames/s", steps=k3, ms_per_step=round(1e3 * el3 / k3, 4),
                                    payload_ok=bool((pm.cpu().numpy() == PAYLOAD[None]).all()),
                                    note="embed+detect on I420 planes (1.5 B/px in, 1.5 B/px out, conversion fused into the kernels); "
                                         "payload read from the WRITTEN planes, i.e. after 4:2:0 subsampling")
        del planes, pout

    # the codec tests/mark.py and tests/detect.py construct, same frames, same step: embed + verify + payloads
    if cfg == 2 and a.codec == "dct" and not a.separate_detect and not a.no_extras:
        e0 = lanes[0]["eng"]

        def svd_step():
            _, c, _ = e0.svd_embed_detect(frames, wm_dev, L=PAYLOAD.size, scale=15, wm_row=rows_dev, out=out)
            return e0.payloads(c, N, perm_dev)
        svd_step()
        fence()
        t0 = time.perf_counter()
        k4 = max(3, min(a.steps, 20))
        for _ in range(k4):
            pm = svd_step()
        fence()
        el4 = time.perf_counter() - t0
        extra["dwtdctsvd"] = dict(value=round(world * n * k4 / el4, 1), unit="frames/s", steps=k4, ms_per_step=round(1e3 * el4 / k4, 4),
                                  payload_ok=bool((pm.cpu().numpy() == PAYLOAD[None]).all()),
                                  algorithmic_GBps=round(n * k4 * 6 * H * W / el4 / 1e9, 1),
                                  note="DwtDctSvd embed + verify + payloads in one 6 B/px pass (scale 15)")
        # the same codec with blk=8 (16x16 pixel tiles, an 8x8 singular-triplet solve per tile; H*W/256 bits per frame)
        def svd8_step():
            _, c, _ = e0.svd_embed_detect(frames, wm_dev, L=PAYLOAD.size, scale=15, wm_row=rows_dev, out=out, blk=8)
            return e0.payloads(c, H * W // 256, perm_dev)
        svd8_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(k4):
            pm = svd8_step()
        fence()
        el5 = time.perf_counter() - t0
        extra["dwtdctsvd_blk8"] = dict(value=round(world * n * k4 / el5, 1), unit="frames/s", steps=k4, ms_per_step=round(1e3 * el5 / k4, 4),
                                       payload_ok=bool((pm.cpu().numpy() == PAYLOAD[None]).all()),
                                       algorithmic_GBps=round(n * k4 * 6 * H * W / el5 / 1e9, 1),
                                       note="DwtDctSvd(blk=8) embed + verify + payloads (16x16 pixel tiles; the tile is read twice, the second time from cache)")

    if rank != 0:
        if world > 1:
            barrier()
            dist.destroy_process_group()
        return

    # what the device is doing under this load: one rocm-smi sample (engine clock, socket power) while the steps keep running --
    # the evidence behind "the socket sits on its power limit" travels with the line
    if cfg == 2 and a.codec == "dct" and world == 1 and not a.no_extras:
        try:
            import re
            import shutil
            import subprocess
            import threading
            smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
            stop = threading.Event()

            def keep_busy():
                torch.cuda.set_device(dev)
                while not stop.is_set():
                    for _ in range(50):
                        hot_path(lanes[0]["eng"], lanes[0]["out"])
                    torch.cuda.synchronize()
            th = threading.Thread(target=keep_busy, daemon=True)
            th.start()
            time.sleep(0.4)
            try:
                txt = subprocess.run([smi, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            finally:
                stop.set()
                th.join(timeout=30)
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            power = re.search(r"Power \(W\): ([0-9.]+)", txt)
            extra["device_under_load"] = dict(sclk_mhz=int(sclk.group(1)) if sclk else None, socket_power_w=float(power.group(1)) if power else None,
                                              note="one rocm-smi sample while embed+detect steps run back to back")
        except Exception as exc:
            extra["device_under_load"] = dict(error=repr(exc))

    # PCIe-inclusive rate (never `value`): frames start and end in pinned host memory, three-stream pipeline
    if cfg == 2 and a.codec == "dct" and world == 1 and not a.no_extras and (H, W) == (1080, 1920):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import pcie_pipeline
            pc = {}
            for fmt in ("rgb24", "i420"):
                f_, g_ = pcie_pipeline.measure(fmt, n=200, B=50, H=H, W=W, eng=lanes[0]["eng"])
                pc[fmt] = dict(frames_per_s=round(f_, 1), GBps_each_way=round(g_, 2))
            extra["pcie_inclusive"] = dict(pc, note="embed+verify with every frame crossing PCIe in and out (pinned host memory, "
                                                    "H2D / kernels / D2H on three streams); tools/pcie_pipeline.py")
        except Exception as exc:
            extra["pcie_inclusive"] = dict(error=repr(exc))
        # the same crossing through the PRODUCT's plugin loop: offmark.video.embedder.Embedder / extractor.Extractor over
        # host frames (what tests/mark.py / tests/detect.py drive), tools/plugin_pipeline_rate.py
        try:
            import plugin_pipeline_rate
            pp = plugin_pipeline_rate.measure(n=800, B=50, H=H, W=W, forms=("rgb24", "yuv420p"))
            pp.update(plugin_pipeline_rate.measure(n=300, B=50, H=H, W=W, forms=("pageable",)))
            extra["plugin_pipeline"] = dict(pp, note="Embedder.start() / Extractor.start() over frames in host memory, three-stream "
                                                     "pipeline inside the plugin classes (offmark/video/pipeline.py): *_pinned = page-locked "
                                                     "reader and writer memory (no host copy), pageable = plain ndarrays in and out (two "
                                                     "threaded host copies per frame)")
        except Exception as exc:
            extra["plugin_pipeline"] = dict(error=repr(exc))

    # achievable HBM bandwidth of this device, same run: 16-byte streaming copy (read + write) and read-only stream
    if frames.numel() >= (1 << 28) and out is not None:
        probe_src, probe_dst = frames, out
    else:
        probe_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 256)
        probe_dst = torch.empty_like(probe_src)
    nbytes = probe_src.numel() // 16 * 16
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    s = _hip.current_stream()

    def probe(fn, moved):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            fn()
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 5 * moved / (e0.elapsed_time(e1) * 1e-3) / 1e9

    copy_gbps = probe(lambda: _hip.check(lib.ofmk_hbm_copy(probe_src.data_ptr(), probe_dst.data_ptr(), nbytes, s)), 2 * nbytes)
    read_gbps = probe(lambda: _hip.check(lib.ofmk_hbm_read(probe_src.data_ptr(), nbytes, sink.data_ptr(), s)), nbytes)

    fps = total_frames * a.steps / elapsed
    frame_bytes = 3 * H * W
    roof = None
    sha = source_sha16()
    if kern:
        # algorithmic bytes per frame and kernel: analyze reads the frame (3 B/px);
        # mark reads it again and writes the marked frame (6 B/px); the fused mark+verify kernel
        # moves the same 6 B/px and spares detect's 3 B/px read.  Sum over a step = 9 B/px.
        alg = {"analyze": frame_bytes, "mark": 2 * frame_bytes, "mark_fused": 2 * frame_bytes, "svd": 2 * frame_bytes}
        names = {"analyze": "analyze_kernel<rgb8>", "mark": "mark_rgb8_kernel", "mark_fused": "mark_rgb8_kernel<fused verify>",
                 "svd": "svd_rgb8_kernel<embed+verify>"}
        ceiling = {"analyze": read_gbps, "mark": copy_gbps, "mark_fused": copy_gbps, "svd": copy_gbps}
        per = {}
        for k, v in kern.items():
            if not v["launches"]:
                continue
            avg_ms = v["ms_total"] / v["launches"]
            # one launch of each kind per chunk and step; the event pool is bounded, so a very long run records its first launches only
            d = dict(avg_launch_ms=round(avg_ms, 5), launches=v["launches"], ms_per_step=round(avg_ms * n_chunks, 4))
            if k in alg:
                frames_per_launch = n / n_chunks
                d["algorithmic_bytes_per_launch"] = int(frames_per_launch * alg[k])
                d["achieved_GBps"] = round(frames_per_launch * alg[k] / (avg_ms * 1e-3) / 1e9, 1)
                d["frac_of_peak"] = round(d["achieved_GBps"] / HBM_PEAK_GBPS, 4)
                d["frac_of_measured_" + ("read" if k == "analyze" else "copy")] = round(d["achieved_GBps"] / ceiling[k], 4)
            per[k] = d
        extra["kernels"] = per
        extra["kernels_note"] = (f"{DOMINANT}: event pairs on its launches in the timed region; the other kinds: a pass of "
                                 f"{max(3, min(a.steps, 20))} steps straight after it with a pair on every launch")
        extra["kernel_ms_per_step"] = round(sum(v["ms_per_step"] for v in per.values()), 4)
        dom = DOMINANT if DOMINANT in per else max((k for k in per if k in alg), key=lambda k: per[k]["ms_per_step"])
        achieved = per[dom]["achieved_GBps"]
        # PMC-measured HBM bytes per launch (separate rocprofv3 passes, tools/prof.sh): quoted only when that profile
        # was taken from exactly these kernel sources and this frame size
        traffic, traffic_src = None, None
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            if tj.get("source_sha16") == sha and (tj["height"], tj["width"]) == (H, W) and dom in tj:
                per_frame = (tj[dom]["fetch_bytes"] + tj[dom]["write_bytes"]) / tj["frames_per_dispatch"]
                traffic = int(per_frame * per[dom]["algorithmic_bytes_per_launch"] / alg[dom])
                traffic_src = f"profiles/{name} (source {sha})"
                break
        roof = dict(bound="hbm", kernel=names[dom], achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=per[dom]["algorithmic_bytes_per_launch"],
                    avg_launch_ms=per[dom]["avg_launch_ms"], launches=per[dom]["launches"],
                    frac_of_measured_copy=round(achieved / copy_gbps, 4),
                    frac_of_measured_read=round(achieved / read_gbps, 4))

    base = None
    if world == 1 and not a.no_cpu_baseline and a.codec == "dct" and mode == "embed_detect":
        try:
            nb = 96 if H * W <= 1920 * 1080 else 24
            wm_cpu = Shuffler(key=0).generate_wm(PAYLOAD, (1, N))
            base = cpu_baseline(synthetic_frames(nb, H, W, seed=2000, device=dev).cpu().numpy(), wm_cpu, a.alpha, a.cpu_seconds)
        except Exception as exc:                       # e.g. no C compiler on the box: report, do not lose the GPU line
            base = dict(value=None, unit="frames/s", cores=0, kind="port", sample=f"cpu baseline failed: {exc!r}")

    # 9 B/px per embed+detect frame with the DCT codec; the DwtDctSvd codec has no frame-global
    # dependency and no separate detect read: 6 B/px; detect only: 3 B/px
    bpp = 3 if mode == "detect" else 9 if a.codec == "dct" else 6
    path_gbps = fps * bpp * H * W / 1e9
    what = {2: "configs[1]", 3: "configs[2]", 4: "configs[3]", 5: "configs[4]"}[cfg]
    op = "embed+detect" if mode == "embed_detect" else "leak detect"
    if cfg in (2, 3):
        workload = f"synthetic {W}x{H} u8 RGB x{n} frames per GPU, "
    else:
        workload = (f"synthetic {W}x{H} u8 RGB, {S} segments x {F} frames sharded over {world} rank(s), "
                    + ("own payload per segment, " if cfg == 4 else f"{C} copies per segment, leak 01201201 + N(0,2) noise, "))
    line = {
        "metric": f"1080p frames/sec {op}" if (H, W) == (1080, 1920) else f"{W}x{H} frames/sec {op}",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload + f"{'DCT' if a.codec == 'dct' else 'DwtDctSvd'} {op}+vote (BASELINE.json {what})",
                   "codec": a.codec, "frames_per_gpu": n, "payload_bits": int(PAYLOAD.size), "alpha": a.alpha,
                   "chunk_frames": chunk,
                   "detect": ("stand-alone kernels" if (a.separate_detect or mode == "detect") else "fused into the mark kernel")
                   if a.codec == "dct" else "fused into the embed kernel",
                   "sharding": f"{'frames' if cfg in (2, 3) else 'segments'}, {world} rank(s), one RCCL all-gather of payloads"},
        "payload_ber": ber, "payload_bit_exact": payload_ok and votes_ok,
        "roofline": roof,
        "path": {"algorithmic_GBps": round(path_gbps, 1), "bytes_per_frame": bpp * H * W,
                 "frac_of_peak": round(path_gbps / (HBM_PEAK_GBPS * world), 4),
                 "frac_of_measured_copy": round(path_gbps / (copy_gbps * world), 4)},
        "hbm_copy_GBps": round(copy_gbps, 1), "hbm_read_GBps": round(read_gbps, 1),
        "source_sha16": sha,
        "collective": {"backend": ("rccl" if a.backend == "nccl" else a.backend) if grouped else None, "ranks": ranks_seen,
                       "self_launched": bool(os.environ.get("OFMK_BENCH_SELF_LAUNCHED"))},
        "rccl_ranks": ranks_seen if (grouped and a.backend == "nccl") else None,
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
