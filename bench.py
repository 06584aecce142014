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
                        --codec dwtdctsvd runs it with the codec the reference's leak scripts construct
                        (tests/detect_watermarks.py:207).

  python bench.py --gpus 1 --steps 100 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...      with N > 1 and no launcher (WORLD_SIZE unset): bench.py starts that same
                                    torch.distributed.run command itself as a child process, before anything
                                    touches the GPU, and relays rank 0's line and the exit code.
  python bench.py --config 4 --emulate-world 8
                                    ONE GPU rehearses rank 0 of an 8-rank job: it processes rank 0's shard only, but
                                    "gathers" (a device copy standing in for the RCCL all-gather) and votes over ALL ranks'
                                    payloads, and also times the whole job on this GPU, so the line's `emulation` object
                                    carries a predicted 8-GPU speed-up: T(whole job, 1 GPU) / T(rank 0's step).  `value`
                                    stays what this one GPU really processed.
`collective.ranks` / `rccl_ranks` = what an all-reduce of ones returned: the ranks the collective library really joined.

Small shards (a 48-frame segment is 0.2 ms of GPU work, less than the host needs to issue a step): `--group G` (default: auto)
issues G steps per host iteration -- the G steps' kernels are replayed as ONE captured hipGraph, their payloads are gathered
and downloaded once, and the host votes on all G steps' payloads in one vectorised call; every step still embeds, detects,
reduces and votes on its own batch (offmark.dist.steps.StepPipeline, which the Runner below specialises).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (config 2: the fused mark+verify kernel:
it re-reads each frame, writes the marked frame and analyzes it).  Launch durations (`kernels`) come
from HIP event pairs the library attaches to every kernel dispatch of the timed steps
(hipExtLaunchKernelGGL start/stop events on the launch stream: the dispatch's own timestamps, no marker
packets).  `mark_order` times the same K steps with the fused mark kernel in both tile orders, interleaved in this
process, next to the policy the timed region ran under (default: the library's static rule on the launch size; no
measurement) and the workgroup -> XCD deal the hardware reported.  `embed_only` / `detect_only`: the two operations
tests/mark.py and tests/detect.py perform, 20 steps each after the timed region (6 and 3 B/px algorithmic).
Set-up also places the two buffers the kernels WRITE -- the engine's workspace and the marked frames' destination -- among candidate
allocations by the real kernels' launch time over the job's frames (offmark/placement.py; `config.placement_probe`; --placement-candidates 1
turns it off): on MI355X their physical backing decides which of three speed levels the frame kernels run at.
`value` is timed after a bounded, reported pre-heat (`config.preheat_ms` of untimed steps: device out of idle); `value_no_preheat` is
the contract read literally -- W warm-up steps from an idle device, then K timed steps -- taken before the pre-heat.
`kernels` has every kernel kind's launch duration with achieved GB/s and fraction of peak, including `mark`, the NON-fused mark
kernel tests/mark.py's operation runs.  `oracle_check`: three frames of the timed batch and what the timed steps wrote for them,
against the C oracle's embed + detect (beside the CPU baseline); an overrun of the parity budgets turns payload_bit_exact false.
`dwtdctsvd.oracle_check` / `dwtdctsvd_blk8.oracle_check`: the same three frames as the codec tests/mark.py constructs marked
them in the side measurement, against the NumPy oracle.
At N > 1 only `value`, `value_no_preheat` and `second_pass` are measured unless --side-measurements is given, and the line gains
`per_rank` (every rank's own ms per step, dominant-kernel and analyze durations: one all-gather of three floats) and
`scaling_efficiency_inputs` (slowest / median / fastest rank); both launch paths give the ranks the same environment
(`collective.env`: rank_environment); an exception on any rank ends the whole job with a non-zero exit and no line (never a rank
left behind in a barrier); a rank without a shard (--segments fewer than the ranks) still joins every collective; and every rank
binds itself to the cores of its GPU's NUMA node before anything touches the GPU (`placement`).
`roofline.traffic` (PMC-measured HBM bytes per launch) is taken from
profiles/ only when that profile was made from exactly the kernel sources that are running (hash stamp),
else null.  `cpu_baseline` is the plain-C restatement of the reference algorithm (oracle/offmark_oracle.c,
bit-identical to the NumPy oracle and the golden vectors; kind "port": OpenCV is not installed, so the
reference itself cannot run, and OpenCV's own float rounding is parity-unpinned) with one OpenMP thread per
frame on the host cores this process may use; `cpu_baseline.variants` adds BASELINE.md's A / B1 / B2 forms (A: ten whole frames
through the reference-shaped per-block loops, one single-threaded worker process per frame).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is achievable
PAYLOAD = np.array([0, 1, 1, 0, 0, 1, 0, 1])
GRAPH_BELOW_BYTES = 100 * 1080 * 1920 * 3      # shards under 100 frames of 1080p: capture the step(s) as a hipGraph


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
    ap.add_argument("--blk", type=int, default=4, choices=[4, 8], help="DwtDctSvd block size (--codec dwtdctsvd)")
    ap.add_argument("--pixfmt", choices=["rgb24", "i420", "nv12"], default="rgb24",
                    help="frame layout in HBM (config 2/3, DCT codec): interleaved rgb24 (the metric) or 4:2:0 planes")
    ap.add_argument("--tile-order", choices=["auto", "xcd", "linear"], default="auto",
                    help="tile order of the frame-writing DCT kernel: auto = the library's static rule on the launch size (default); xcd / linear force one")
    ap.add_argument("--placement-candidates", type=int, default=8,
                    help="set-up: the engine's workspace and the output buffer are picked among this many candidate allocations by the real "
                         "kernels' launch time (offmark/placement.py; reported as config.placement_probe); 0 or 1 = first allocation, no probe")
    ap.add_argument("--segments", type=int, default=8, help="configs 4/5: segments of the job (8 = BASELINE.json's; fewer than the ranks "
                                                            "leaves ranks without a shard: rehearsals and tests)")
    ap.add_argument("--preheat-ms", type=float, default=250.0,
                    help="untimed steps of the workload for about this long in set-up, before the W warm-up steps (device out of idle; 0 = none)")
    ap.add_argument("--side-measurements", action="store_true",
                    help="N > 1: also run the side measurements (default there: value and second_pass only)")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the rank to its GPU's NUMA-local cores")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one GPU rehearses rank 0 of an M-rank job (see module text); needs --gpus 1")
    ap.add_argument("--group", type=int, default=0, help="steps issued per host iteration (0 = auto: 1 for shards of >= 100 "
                                                         "1080p frames, else as many as make ~300 frames)")
    ap.add_argument("--no-graph", action="store_true", help="never capture the step as a hipGraph")
    ap.add_argument("--graph", action="store_true", help="capture the step(s) as a hipGraph whatever the shard size (experiments)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (a one-GPU box cannot run RCCL across ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not attach HIP events to the kernel launches (roofline becomes null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements")
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
    """The oracle on the host cores, embed+detect on a bounded sample of the same workload (SURVEY 8d / BASELINE.md 3).
    value: the C restatement (oracle/offmark_oracle.c, bit-identical to the NumPy oracle), one OpenMP thread per
    frame on every core this process may use.  variants: BASELINE.md section 3's forms of the NumPy oracle --
      A   reference-shaped per-block Python loops, TEN whole frames, one single-threaded worker process per frame side by side
          on the host cores (~11 s of wall); the per-core rate is reported -- the reference is single-threaded;
      B1  all-blocks-at-once NumPy on one core, 100 frames if they fit ~25 s, else as many as do (count stated);
      B2  B1 in one worker process per core, >= 100 frames in all (oracle/cpu_baseline_worker.py; separate processes that
          never touch the GPU)."""
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
    small = H * W < 1920 * 1080                       # shortened test runs: keep the variants proportionate
    try:
        # B1: vectorised NumPy, one core
        enc = orc.DctEncoderOracle(alpha=alpha)
        enc.read_wm(wm)
        t1, k, want, limit = time.perf_counter(), 0, (10 if small else 100), (5.0 if small else 25.0)
        b1_ok = True
        while k < want:
            b = orc.check_frame(orc.mark_frame(frames_u8[k % n], enc), orc.DctDecoderOracle(alpha=alpha))
            b1_ok &= bool(np.array_equal(orc.deshuffle(b, PAYLOAD.size, 0), PAYLOAD))
            k += 1
            spent = time.perf_counter() - t1
            if k >= 3 and spent * (k + 1) / k > limit:
                break
        variants["B1_numpy_vectorised"] = dict(value=round(k / (time.perf_counter() - t1), 2), unit="frames/s", cores=1, frames=k,
                                               payload_ok=b1_ok, note=None if k >= want else f"{want} frames do not fit {limit:.0f} s on one core")
    except Exception as exc:
        variants["error_B1"] = repr(exc)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
    # A: the reference-shaped per-block Python loops (dct_encoder.py:18-102, dct_decoder.py:10-27), TEN whole frames (SURVEY 8d's
    # count), each on one core: one single-threaded worker process per frame, side by side on the idle host cores, so the ten
    # frames cost the wall time of one (or of ceil(10 / cores)); the rate reported is PER CORE, which is what the reference is
    try:
        workers_a = min(10, threads)
        per_a = -(-10 // workers_a)
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline_worker.py"), str(H), str(W),
                                   str(per_a), str(5000 + 16 * i), str(alpha), "loop"], stdout=subprocess.PIPE, text=True, env=env)
                 for i in range(workers_a)]
        spans = [json.loads(p.communicate(timeout=1200)[0].strip().splitlines()[-1]) for p in procs]
        per_core = [per_a / (s_["t1"] - s_["t0"]) for s_ in spans]
        wall_a = max(s_["t1"] for s_ in spans) - min(s_["t0"] for s_ in spans)
        variants["A_reference_shaped_loop"] = dict(value=round(float(np.mean(per_core)), 4), unit="frames/s", cores=1, frames=workers_a * per_a,
                                                   payload_ok=all(s_["ok"] for s_ in spans), workers=workers_a, wall_s=round(wall_a, 1),
                                                   note=f"{workers_a * per_a} whole {W}x{H} frames through the reference-shaped per-block loops, one "
                                                        f"single-threaded worker process per {'frame' if per_a == 1 else str(per_a) + ' frames'}; value = mean "
                                                        f"per-core rate (min {min(per_core):.4f}, max {max(per_core):.4f}); the reference is single-threaded "
                                                        "(video/embedder.py:19-27)")
    except Exception as exc:
        variants["A_reference_shaped_loop"] = dict(value=None, error=repr(exc))
    # B2: B1 fanned out, one worker process per core, >= 100 frames in all
    try:
        per = max(2, -(-(10 if small else 100) // threads))
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline_worker.py"), str(H), str(W),
                                   str(per), str(3000 + i), str(alpha)], stdout=subprocess.PIPE, text=True, env=env)
                 for i in range(threads)]
        spans = [json.loads(p.communicate(timeout=600)[0].strip().splitlines()[-1]) for p in procs]
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


def oracle_check(frames_u8, marked_u8, gpu_bits_of_oracle_marked, wm_rows, alpha, payloads_gpu):
    """Oracle check of the TIMED workload itself ("identical to the oracle's output"; dct_decoder.py:10-27): a few
    frames of the timed batch and the marked frames the timed steps wrote for them, against the C restatement's embed + detect
    of the same frames (oracle/offmark_oracle.c, bit-identical to the NumPy oracle and the golden vectors).  Runs beside the CPU
    baseline, after the timed region.  Budgets = the parity tests' (tests/test_gpu_parity.py): marked pixels <= 1 LSB on <= 1e-5
    of the samples over sign-determined blocks (|C21| > 1e-3 in the oracle; elsewhere the sign of a ~1e-6 coefficient decides a
    whole quantisation step and no independent implementation can reproduce it); raw bits <= 1e-4 of the blocks; payloads equal.
    gpu_bits_of_oracle_marked(ref_marked) -> the GPU detector's raw bits [k, N] of the ORACLE's marked frames (same input to
    both detectors)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    import offmark_oracle as orc
    k, H, W, _ = frames_u8.shape
    nblk = (H // 8) * (W // 8)
    px_bad = px_n = px_max = amb = bits_bad = 0
    payload_equal = True
    ref_marked = np.empty_like(frames_u8)
    ref_bits = []
    for i in range(k):
        wm = wm_rows[i]
        ref_marked[i] = c_oracle.mark_frames(frames_u8[i:i + 1], wm, alpha=alpha, legacy=True, threads=1)[0][0]
        ref_bits.append(c_oracle.check_frames(ref_marked[i:i + 1], alpha=alpha, legacy=True, threads=1)[0][0])
        enc = orc.DctEncoderOracle(alpha=alpha)
        enc.read_wm(np.asarray(wm).reshape(1, -1))            # read_wm keeps wm[0], like the reference (dct_encoder.py:10-11)
        enc.encode(orc.bgr2yuv_f32(frames_u8[i].astype(np.float32)))
        ok = np.abs(enc.debug["c21_pre"]) > 1e-3
        amb += int((~ok).sum())
        mask = np.zeros((H, W), bool)
        mask[: ok.shape[0] * 8, : ok.shape[1] * 8] = np.kron(ok, np.ones((8, 8), bool))
        d = np.abs(marked_u8[i].astype(np.int16) - ref_marked[i].astype(np.int16))[mask]
        px_bad += int((d > 0).sum())
        px_n += int(d.size)
        px_max = max(px_max, int(d.max()) if d.size else 0)
        payload_equal &= bool(np.array_equal(orc.deshuffle(ref_bits[-1], PAYLOAD.size, 0), payloads_gpu[i]))
    got = gpu_bits_of_oracle_marked(ref_marked)
    for i in range(k):
        bits_bad += int((got[i].reshape(-1)[:nblk] != ref_bits[i].reshape(-1)[:nblk]).sum())
    within = (px_max <= 1 and px_bad <= max(1, int(px_n * 1e-5)) and bits_bad <= max(1, int(k * nblk * 1e-4)) and payload_equal)
    return dict(frames=k, pixels_compared=px_n, pixels_differing_over_determined_blocks=px_bad, max_pixel_difference=px_max,
                sign_ambiguous_blocks=amb, blocks=k * nblk, raw_bits_compared=k * nblk, raw_bits_differing=bits_bad,
                payload_equal=payload_equal, within_budget=bool(within),
                note="frames of the timed batch and the marked frames the timed steps wrote, against the C oracle's embed + detect of the same "
                     "frames; raw bits: the GPU detector and the oracle's on the ORACLE's marked frames; budgets: <= 1 LSB on <= 1e-5 of the "
                     "samples over sign-determined blocks, <= 1e-4 of the raw bits, payloads equal")


def oracle_check_svd(frames_u8, marked_u8, gpu_bits_of_oracle_marked, wm_rows, payloads_gpu, blk, scale=15.0):
    """The same check for the DwtDctSvd codec -- what tests/mark.py and tests/detect.py construct
    (embed/dwt_dct_svd_encoder.py:19-45, extract/dwt_dct_svd_decoder.py:12-37) -- against the NumPy oracle (no C restatement of this
    codec exists).  Budgets = tests/test_gpu_svd.py's: marked pixels <= 1 LSB on <= 2e-5 of the samples over DETERMINED tiles (a
    tile is left out when its top singular value lies within 1e-3 of a multiple of the quantisation step, where the floor
    division flips on the last float bits, or when its two largest singular values coincide to 1e-3, where the rank-1 direction is
    not defined); raw bits <= 1e-4 of the tiles; payloads equal."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import offmark_oracle as orc
    k, H, W, _ = frames_u8.shape
    px = 2 * blk
    px_bad = px_n = px_max = left_out = bits_bad = tiles = 0
    payload_equal = True
    ref_marked = np.empty_like(frames_u8)
    ref_bits = []
    for i in range(k):
        enc = orc.DwtDctSvdEncoderOracle(scales=(0.0, scale, 0.0), blk=blk)
        enc.read_wm(np.asarray(wm_rows[i]).reshape(1, -1))
        ref_marked[i] = orc.mark_frame(frames_u8[i], enc)
        dbg = enc.debug_ch[1]
        s0, gap = dbg["s0"].astype(np.float64), dbg["gap"]
        frac = np.mod(s0, scale)
        ok = (np.minimum(frac, scale - frac) > 1e-3 * np.maximum(1.0, s0 / 100)) & (gap < 1 - 1e-3)
        left_out += int((~ok).sum())
        tiles += int(ok.size)
        mask = np.zeros((H, W), bool)
        mask[: ok.shape[0] * px, : ok.shape[1] * px] = np.kron(ok, np.ones((px, px), bool))
        d = np.abs(marked_u8[i].astype(np.int16) - ref_marked[i].astype(np.int16))[mask]
        px_bad += int((d > 0).sum())
        px_n += int(d.size)
        px_max = max(px_max, int(d.max()) if d.size else 0)
        ref_bits.append(orc.check_frame(ref_marked[i], orc.DwtDctSvdDecoderOracle(scales=(0.0, scale, 0.0), blk=blk)).reshape(-1))
        payload_equal &= bool(np.array_equal(orc.deshuffle(ref_bits[-1], PAYLOAD.size, 0), payloads_gpu[i]))
    got = gpu_bits_of_oracle_marked(ref_marked)
    for i in range(k):
        bits_bad += int((got[i].reshape(-1) != ref_bits[i]).sum())
    within = px_max <= 1 and px_bad <= max(1, int(px_n * 2e-5)) and bits_bad <= max(1, int(tiles * 1e-4)) and payload_equal
    return dict(frames=k, pixels_compared=px_n, pixels_differing_over_determined_tiles=px_bad, max_pixel_difference=px_max,
                tiles=tiles, tiles_left_out=left_out, raw_bits_differing=bits_bad, payload_equal=payload_equal, within_budget=bool(within))


def attack_suite(torch, detect, clean, per_seg, payloads, chosen, fp, vote_segments, deg, n_bits, H, W, codec):
    """BASELINE.json configs[4]: the leak's frames under the build-defined attacks of SURVEY 8d (none exist upstream: the
    reference's only lossy leg is a JPEG, tests/test.py:99, and its HLS re-encode).  `clean`: marked frames [S * per_seg, H, W, 3]
    on the device; `detect(frames) -> counts [n, L]` is the codec's read-out.  Per attack: payload bit error rate over the frames,
    frames decoded exactly, segments whose Counter vote is right, and whether the leak's copy sequence comes out.  Reported as
    measured: cropping moves the 8x8 grid and is EXPECTED to defeat a block-transform QIM scheme (so is a strong re-quantisation
    such as JPEG quality 75); a mild rescale may or may not survive depending on the content; only "none" and "noise" are
    parity-gated (tests)."""
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
            counts = detect(fn(clean))
            got = deg.degenerate_counts(counts.cpu().numpy(), n_bits)
            votes = vote_segments(got, seg)
            seg_ok = sum(int(v[0] is not None and np.array_equal(v[0], payloads[s])) for s, v in votes.items())
            out[name] = dict(payload_ber=round(float((got != want).mean()), 4), frames_exact=round(float((got == want).all(axis=1).mean()), 4),
                             segments_ok=f"{seg_ok}/{S}", copies_recovered=fp.identify_copies({s + 1: v for s, v in votes.items()}) == chosen)
        except Exception as exc:                                   # never lose the line over a side report
            out[name] = dict(error=repr(exc))
    out["note"] = (f"{per_seg} frames per segment, {codec} codec; attacks are build-defined tensor ops between embed and detect (Pillow for the "
                   "JPEGs); a crop moves the block grid and is expected to fail for this scheme; the others are reported as measured")
    return out


def plugin_yuv32f_rates(H, W, alpha, frame_u8):
    """The LITERAL plugin boundary an unmodified reference Embedder / Extractor would call once per frame
    (src/offmark/video/embedder.py:35, extractor.py:32; tests/test.py:91-118): encode(yuv) / decode(yuv) on ONE host float32
    YUV frame -- upload 12 B/px, kernels, download.  Frames/s per call, both codecs; not tuned (a single-frame API), reported
    so that path has a number."""
    from offmark.embed.dct_encoder import DctEncoder
    from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder
    from offmark.extract.dct_decoder import DctDecoder
    from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.generator.shuffler import Shuffler
    from offmark.video.color import bgr2yuv
    yuv0 = bgr2yuv(frame_u8.astype(np.float32))
    deg = DeShuffler(key=0).set_shape(PAYLOAD.shape)
    out = {}
    for name, enc, dec in (("dct", DctEncoder(alpha=alpha), DctDecoder(alpha=alpha)),
                           ("dwtdctsvd", DwtDctSvdEncoder(), DwtDctSvdDecoder())):
        enc.read_wm(Shuffler(key=0).generate_wm(PAYLOAD, enc.wm_capacity((H, W, 3))))
        marked = enc.encode(yuv0.copy())
        bits = dec.decode(marked)                             # warm: allocations, code objects
        k = 8
        copies = [yuv0.copy() for _ in range(k)]              # the caller's frames exist before the call: not part of its time
        t0 = time.perf_counter()
        for c_ in copies:
            marked = enc.encode(c_)
        t1 = time.perf_counter()
        for _ in range(k):
            bits = dec.decode(marked)
        t2 = time.perf_counter()
        out[name] = dict(encode_fps=round(k / (t1 - t0), 1), decode_fps=round(k / (t2 - t1), 1),
                         encode_decode_fps=round(k / (t2 - t0), 1), calls=k,
                         payload_ok=bool(np.array_equal(deg.degenerate(bits), PAYLOAD)))
    out["note"] = (f"DctEncoder.encode(yuv) / DctDecoder.decode(yuv) and the DwtDctSvd pair on one host float32 {W}x{H} YUV frame per call "
                   "(pageable ndarray up, kernels, the frame down into the caller's array): the path an unmodified reference "
                   "Embedder takes (video/embedder.py:35).  The batched u8 path is the product's fast path")
    return out


class stdout_to_stderr:
    """File-descriptor-level redirection of stdout into stderr for the duration of a block: native libraries print there
    (RCCL writes a five-line version banner to STDOUT when its first communicator comes up), and this program's stdout
    carries exactly one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def _cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out.update(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes():
    """NUMA node of every AMD GPU of this host in PCI bus order -- the order the HIP runtime enumerates them in -- from sysfs
    (/sys/class/drm/card*/device/numa_node); -1 where the platform reports none."""
    import glob
    import re
    found = []
    for dev in glob.glob("/sys/class/drm/card*/device"):
        if not re.fullmatch(r"card\d+", os.path.basename(os.path.dirname(dev))):
            continue                                                     # connectors (card0-DP-1) are not devices
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            found.append((os.path.basename(os.path.realpath(dev)), int(open(os.path.join(dev, "numa_node")).read())))
        except (OSError, ValueError):
            continue
    return [node for _, node in sorted(found)]


def bind_to_gpu_numa(gpu_of_rank, local_rank):
    """Bind this process to the cores of its GPU's NUMA node BEFORE anything touches the GPU (the runtime's helper threads
    inherit the mask).  Ranks whose GPUs share a node split that node's cores between them, so eight ranks on one host do not
    issue their launches from each other's cores (the shard shape is tests/segment_mark_detect_hls.py:407-412,
    one independent worker per segment).  gpu_of_rank: HIP device index of every local rank.  Never fatal: a host that does
    not expose the topology leaves the process where the launcher put it."""
    info = dict(bound=False)
    try:
        nodes = gpu_numa_nodes()
        vis = None
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):       # device i of the process = entry i of the list
            text = os.environ.get(var, "")
            if text and all(x.strip().isdigit() for x in text.split(",")):
                ids = [int(x) for x in text.split(",")]
                vis = ids if vis is None else [vis[i] for i in ids]
        phys = [(vis[g] if vis else g) for g in gpu_of_rank]
        node = nodes[phys[local_rank]]
        info.update(gpu=phys[local_rank], numa_node=node, gpus_seen=len(nodes))
        if node < 0:
            info["note"] = "the GPU reports no NUMA node"
            return info
        allowed = os.sched_getaffinity(0)
        cpus = sorted(_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) & allowed)
        peers = [r for r in range(len(phys)) if nodes[phys[r]] == node]
        per = len(cpus) // max(len(peers), 1)
        mine = cpus[peers.index(local_rank) * per:(peers.index(local_rank) + 1) * per] if per >= 2 else cpus
        if not mine:
            info["note"] = "no core of that node in this process's affinity mask"
            return info
        os.sched_setaffinity(0, mine)
        runs, start = [], mine[0]                                        # compact cpulist form: 0-63,128-191
        for a_, b_ in zip(mine, mine[1:] + [None]):
            if b_ != a_ + 1:
                runs.append(f"{start}-{a_}" if a_ != start else str(a_))
                start = b_
        info.update(bound=True, cpus=",".join(runs), n_cpus=len(mine), ranks_on_node=len(peers))
    except Exception as exc:
        info["error"] = repr(exc)
    return info


def inject_failure(where, rank):
    """Test hook (tests/test_gpu_parity.py): OFMK_BENCH_INJECT_FAILURE="<rank>:<where>" raises in that rank at that point."""
    want = os.environ.get("OFMK_BENCH_INJECT_FAILURE", "")
    if want and want == f"{rank}:{where}":
        raise RuntimeError(f"injected failure in rank {rank} at {where}")


# What a rank needs in its environment before the runtime loads, whichever way it was started.  HSA_ENABLE_IPC_MODE_LEGACY=0:
# this GPU pool's host driver supports only dmabuf IPC; with the legacy mode RCCL's (and torch's) cross-process buffer sharing
# fails with "hipIpcGetMemHandle: invalid argument" (the pool's environment notes: the image exports the variable for that
# reason; a driver-started `python -m torch.distributed.run ... bench.py` inherits it, but nothing guarantees an outside
# launcher's environment, so every rank also sets it for itself -- setdefault: an explicit choice of the caller stands).
RANK_ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def rank_environment(env, world):
    """Apply RANK_ENV to `env` (os.environ of a rank, or the environment of the child launcher) when the job has more than one
    rank.  Both launch paths -- bench.py starting its own ranks, and an outside torch.distributed.run -- go through this, so they
    see the same environment (tests/test_dist_gloo.py).  Returns what the job's ranks end up with."""
    if world > 1:
        for k, v in RANK_ENV.items():
            env.setdefault(k, v)
    return {k: env.get(k) for k in RANK_ENV}


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
    rank_environment(env, a.gpus)
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
    if a.emulate_world and a.gpus != 1:
        raise SystemExit("--emulate-world needs --gpus 1 (it rehearses rank 0 of an M-rank job on ONE GPU)")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    # environment and placement first: nothing has touched the GPU yet (no torch import, no HIP call)
    env_world, env_local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    rank_env = rank_environment(os.environ, env_world)
    n_local = int(os.environ.get("LOCAL_WORLD_SIZE", env_world))
    affinity_at_start = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    placement = dict(bound=False, note="--no-bind") if a.no_bind else \
        bind_to_gpu_numa([0] * n_local if a.single_device else list(range(n_local)), env_local)
    import torch
    import torch.distributed as dist
    from offmark import _hip
    from offmark import engine as engine_mod
    from offmark import fingerprint as fp
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.steps import StepPipeline
    from offmark.dist.vote import gather_payloads, init_from_env, shard_range, vote_segments
    from offmark.engine import DctEngine, balanced_chunk, default_chunk_frames
    from offmark.generator.shuffler import Shuffler
    from offmark.synthetic import synthetic_frames

    if a.single_device:
        os.environ["LOCAL_RANK"] = "0"
    with stdout_to_stderr():
        rank, world = init_from_env(a.backend, force=a.rehearse_collectives)
    grouped = world > 1 or a.rehearse_collectives          # a process group exists: run the collectives
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    emu = a.emulate_world if a.emulate_world > 1 else 0
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = _hip.load()
    flags = _hip.F_SEPARATE_DETECT if a.separate_detect else 0
    ranks_seen = 1
    if grouped:                                   # how many ranks the collective library really connected: sum of ones
        with stdout_to_stderr():                  # the first collective brings the communicator up (RCCL prints its banner here)
            one = torch.ones(1, dtype=torch.int32, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(one)
            ranks_seen = int(one.item())
            torch.cuda.synchronize()

    cfg = a.config
    H = a.height or (2160 if cfg == 3 else 1080)
    W = a.width or (3840 if cfg == 3 else 1920)
    N = H * W // 64
    L = int(PAYLOAD.size)
    planar = a.pixfmt != "rgb24"
    if planar and (cfg not in (2, 3) or a.codec != "dct" or a.separate_detect or H % 8 or W % 8):
        raise SystemExit("--pixfmt i420/nv12: configs 2/3, DCT codec, fused verify, H and W multiples of 8")
    # bits in the decoder's vector: what the degenerator's means divide by (dwt_dct_svd_decoder.py:14 for blk = 8)
    n_bits = DctEngine.svd_bits_per_frame(H, W, a.blk) if a.codec == "dwtdctsvd" else N
    deg = DeShuffler(key=0).set_shape(PAYLOAD.shape)
    perm_dev = torch.as_tensor(deg.payload_idx, dtype=torch.int32).to(dev)
    S, F, C = a.segments, a.frames or 48, 3                  # configs 4/5: segments, frames per segment, copies

    def barrier():
        if a.backend == "nccl":
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()

    # ---- workload: one rank's share of a BASELINE config --------------------------------------------------------
    def make_job(shard_rank, shard_world, share=None):
        """The frames, watermarks and bookkeeping of rank `shard_rank` of `shard_world` ranks.  `share`: a job of the WHOLE
        workload (one rank) whose tensors this shard may slice instead of generating its own (emulation: both live on this GPU)."""
        j = SimpleNamespace(chosen=None, leak_sample=None, keep=0, rows_dev=None, wm_dev=None, planes=None)
        if cfg in (2, 3):
            j.scaling, j.equal = "weak", True
            j.n = a.frames or (1000 if cfg == 3 else 300)
            j.frames = share.frames if share is not None else synthetic_frames(j.n, H, W, seed=2000 + shard_rank, device=dev)
            j.wm_table = Shuffler(key=0).generate_wm(PAYLOAD, (1, N)).astype(np.uint8)
            j.seg_global = np.repeat(np.arange(shard_world), j.n)          # one "segment" per rank
            j.first = shard_rank * j.n
            j.expected = {r: PAYLOAD for r in range(shard_world)}
            j.total_frames = shard_world * j.n
            j.mode = "embed_detect"
            j.wm_dev = torch.from_numpy(j.wm_table).to(dev)
            if planar:
                j.planes = share.planes if share is not None else DctEngine(device=dev).rgb_to_yuv420(j.frames, a.pixfmt)
        else:
            j.scaling = "strong"                                 # the 8-segment job is split over the ranks
            s0, s1 = shard_range(S, shard_rank, shard_world)
            j.n, j.first = (s1 - s0) * F, s0 * F
            j.equal = S % shard_world == 0
            j.seg_global = np.repeat(np.arange(S), F)
            j.total_frames = S * F
            if share is not None:
                src = share.src[j.first:j.first + j.n]
            else:
                # every rank's frames come from ONE generator stream (seed 4000), so that a shard is the same frames whether
                # it is generated alone or sliced out of the whole job
                src = synthetic_frames(S * F, H, W, seed=4000, device=dev)[j.first:j.first + j.n].clone() if shard_world > 1 \
                    else synthetic_frames(S * F, H, W, seed=4000, device=dev)
            j.src = src
            if cfg == 4:
                # segment s carries format(s % 256, '08b'); numbering starts at 1 because segment 0's all-zero payload
                # cannot be decoded by the reference's mid-range threshold (de_shuffler.py:20-21)
                payloads = np.stack([fp.payload_for_segment(s + 1) for s in range(S)])
                j.wm_table = np.stack([Shuffler(key=0).generate_wm(p, (1, N))[0] for p in payloads]).astype(np.uint8)
                j.rows_dev = torch.from_numpy(np.repeat(np.arange(s0, s1), F).astype(np.int32)).to(dev)
                j.wm_dev = torch.from_numpy(j.wm_table).to(dev)
                j.expected = {s: payloads[s] for s in range(S)}
                j.frames = src
                j.mode = "embed_detect"
            else:
                j.chosen = fp.select_copies(("01201201" * (S // 8 + 1))[:S], S, C)
                j.expected = {s: fp.payload_for_segment(s + 1, j.chosen[s]) for s in range(S)}
                j.mode = "detect"
                if share is not None:
                    j.frames = share.frames[j.first:j.first + j.n]
                else:
                    table = np.stack([Shuffler(key=0).generate_wm(fp.payload_for_segment(s + 1, c), (1, N))[0]
                                      for s in range(S) for c in range(C)]).astype(np.uint8)
                    rows = np.array([(s * C + j.chosen[s]) for s in range(s0, s1) for _ in range(F)], dtype=np.int32)
                    setup = DctEngine(device=dev)
                    if j.n == 0:
                        leak = src
                    elif a.codec == "dct":
                        leak = setup.embed(src, table, alpha=a.alpha, wm_row=rows)
                    else:
                        leak = setup.svd_embed(src, table, scale=15, wm_row=rows, blk=a.blk)
                    g = torch.Generator(device=dev).manual_seed(7 + shard_rank)    # build-defined attack (i): N(0, 2) + round/clip
                    j.frames = (leak.float() + 2.0 * torch.randn(leak.shape, device=dev, generator=g)).round().clamp(0, 255).to(torch.uint8) \
                        if j.n else leak
                    # a few clean frames of every local segment for the attack suite reported next to the line (not timed)
                    j.keep = min(F, 8)
                    j.leak_sample = leak.view(s1 - s0, F, H, W, 3)[:, :j.keep].reshape(-1, H, W, 3).clone() if j.n else None
                    del leak, setup
        j.shard_world, j.shard_rank = shard_world, shard_rank
        # the payload every frame of the WHOLE job should decode to (what the other ranks contribute in an emulation)
        j.expected_rows = np.stack([j.expected[s] for s in j.seg_global]).astype(np.uint8)
        return j

    if emu:
        if cfg in (4, 5) and S % emu:
            raise SystemExit(f"--emulate-world {emu}: the {S} segments must split evenly over the emulated ranks")
        job_full = make_job(0, 1)
        job = make_job(0, emu, share=job_full)
    else:
        job_full = None
        job = make_job(rank, world)
    n = job.n
    mode = job.mode
    DOMINANT = ("analyze" if mode == "detect" else "mark" if a.separate_detect else "mark_fused") if a.codec == "dct" else "svd"
    if planar:
        DOMINANT = "planar_mark"
    opts_plain = _hip.Opts(flags, 0, None)

    # ---- the step over one job ---------------------------------------------------------------------------------
    class Runner(StepPipeline):
        """offmark.dist.steps.StepPipeline (lanes, grouped hipGraph replay on two branches, side-stream gather + download, host vote)
        over one job of this benchmark: what a step is, the emulated gather, the contract's fences and the checks."""

        def __init__(self, j, n_lanes=1, group=1, graph=False, emulate=False):
            self.j, self.emulate = j, emulate
            self.chunk = a.chunk or default_chunk_frames(H, W)
            self.cf = balanced_chunk(max(j.n, 1), self.chunk)
            self.n_chunks = max(1, -(-j.n // self.cf))
            src = j.planes if planar else j.frames
            super().__init__(dev, j.n, L, j.seg_global,
                             make_engine=lambda: DctEngine(device=dev, chunk_frames=self.chunk, opts=opts_plain, tile_order=a.tile_order),
                             make_out=lambda: torch.empty_like(src) if j.mode == "embed_detect" else None,
                             issue=self.issue_step, lanes=n_lanes, group=group, graph=graph, gather=self.gather_rows, equal_shards=j.equal)
            if emulate:        # the gathered buffer of an M-rank job, rank-major [M, G, n, L]: the other ranks' parts are pre-filled
                ew = j.shard_world
                per_rank = torch.from_numpy(j.expected_rows).to(dev).view(ew, 1, j.n, L).expand(ew, self.G, j.n, L).contiguous()
                self.everyone = [per_rank.clone() for _ in range(2)]

        def set_opts(self, o):
            for e in self.engines():
                e.opts = o

        def set_order(self, order):
            for e in self.engines():
                e._order_mode = order
            self.drop_graphs()

        def issue_step(self, e, out, slot):
            """embed + detect (config 5: detect only) + per-frame payloads for this rank's frames -> slot [n, L] uint8 on the device."""
            j = self.j
            if j.mode == "detect":
                if a.codec == "dct":
                    counts, _ = e.detect(j.frames, L, alpha=a.alpha)
                else:
                    counts, _ = e.svd_detect(j.frames, L, scale=15, blk=a.blk, partial=True)       # per-workgroup sums: no fill dispatch
            elif planar:
                _, counts, _ = e.embed_detect_yuv420(j.planes, H, W, j.wm_dev, L, alpha=a.alpha, out=out, layout=a.pixfmt)
            elif a.codec == "dct":
                _, counts, _ = e.embed_detect(j.frames, j.wm_dev, L=L, alpha=a.alpha, wm_row=j.rows_dev, out=out)
            else:
                _, counts, _ = e.svd_embed_detect(j.frames, j.wm_dev, L=L, scale=15, wm_row=j.rows_dev, out=out, blk=a.blk, partial=True)
            e.payloads(counts, n_bits, perm_dev, out=slot)

        def hot_path(self, lane):
            self.step(lane)

        def gather_rows(self, mine, g, size):
            if self.emulate:                                               # a device copy where the RCCL all-gather would be
                buf = self.everyone[g & 1]
                buf[0, :size].copy_(mine.view(size, self.j.n, L))
                return buf[:, :size].reshape(-1, L)
            if a.backend == "gloo" and grouped:                            # rehearsal: gloo gathers host tensors
                return gather_payloads(mine.cpu(), equal_shards=self.j.equal, force=grouped)
            return gather_payloads(mine, equal_shards=self.j.equal, force=grouped)        # RCCL all-gather (N > 1)

        def prepare(self):
            """One-time set-up, not a workload step: allocate the scratch for the chunk size in use, let the runtime load the code
            objects (one full-size pass, so that profiles only ever see full-size launches),
            exercise the download path, capture the G-step graphs when asked.  Even --warmup 0 then times steady-state steps."""
            if self.j.n:
                for e in self.engines():
                    e.workspace(H, W, e._chunk(self.j.n, H, W))
            self.place()
            super().prepare()

        placement = dict(candidates=1, note="off")

        def place(self):
            """Where the buffers the kernels WRITE live -- the engine's workspace (the records) and the marked frames' destination --
            decides which of three speed levels the frame kernels run at; both are this side's to allocate, so they are
            picked once, at set-up, among a few candidate allocations by the real kernels' launch time over the job's own frames
            (offmark/placement.py).  Not a workload step; every rank for itself; before any graph is captured."""
            j = self.j
            if not (a.placement_candidates > 1 and j.n and not planar and a.codec == "dct" and j.frames.is_contiguous()):
                return
            for lane in self.lanes:
                for eng_name, out_name in (("eng", "out"), ("eng2", "out2")):
                    e = getattr(lane, eng_name)
                    if e is None:
                        continue
                    want_out = j.mode == "embed_detect"
                    if want_out:
                        setattr(lane, out_name, None)                      # the buffer make_out() made goes back to the allocator first
                    out, rep = e.place_buffers(j.frames, want_out=want_out, candidates=a.placement_candidates)
                    if want_out:
                        setattr(lane, out_name, out)
                    if lane is self.lanes[0] and eng_name == "eng":
                        self.placement = rep

        def preheat(self, ms):
            """Bring the device from idle to its operating state before the contract's W warm-up steps: untimed steps of the
            workload itself for about `ms` milliseconds (--preheat-ms, reported as config.preheat_ms; 0 = none).  A device
            coming out of idle keeps speeding up for ~100 ms of load (profiles/r4_idle_gap.txt: first launches 0.79 ms, settled
            0.70), and W = 5 warm-up steps are 6 ms of it: without this a short timed region measures the ramp, not the path
            (20 steps from cold: 260 k frames/s, the same 20 steps straight after: 274 k).  Rounds 1-4 had it implicitly (round 4's
            tile-order calibration in set-up was 279 ms of load); now it is explicit, bounded and in the line.  Not a workload
            step: nothing here is timed or counted, and every rank does the same (grouped steps keep their collectives in step)."""
            if ms <= 0:               # (a rank WITHOUT a shard still takes part: its steps issue no kernels but do gather, and it joins the
                return 0.0            # stop vote below -- returning early here left its peers alone in their collectives: ADVICE r5)
            t0 = time.perf_counter()
            chunk_steps = max(self.G, self.G * max(1, int(8 // max(self.G, 1))))          # ~8 steps between host checks, whole groups
            while True:
                self.run(chunk_steps)
                torch.cuda.synchronize()
                go = 1e3 * (time.perf_counter() - t0) < ms
                if grouped:          # every step gathers over the ranks, so the ranks must stop after the SAME chunk: the first rank
                    flag = torch.tensor([1.0 if go else 0.0], device=dev if a.backend == "nccl" else "cpu")      # whose clock says so decides for all
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    go = float(flag.item()) != 0.0
                if not go:
                    break
            return round(1e3 * (time.perf_counter() - t0), 1)

        def fence(self):
            torch.cuda.synchronize()
            if grouped:
                barrier()
            torch.cuda.synchronize()

        def timed(self, steps):
            self.fence()
            self.host_s.update(enqueue=0.0, vote=0.0)
            t0 = time.perf_counter()
            votes, size = self.run(steps)
            self.fence()
            el = time.perf_counter() - t0
            self.own_elapsed = el                      # this rank's own span (per_rank in the line)
            if grouped:
                t = torch.tensor([el], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el, votes, size

        def votes_ok(self, votes, size):
            """every step of the (last) group: every segment's vote equals its payload (config 5: and the leak's copy sequence)."""
            j = self.j
            ok = len(votes) == size * len(j.expected)
            for g_ in range(size):
                vg = {s: votes.get(g_ * self.S_ids + s if size > 1 else s) for s in j.expected}
                ok = ok and all(v is not None and v[0] is not None and np.array_equal(v[0], j.expected[s]) for s, v in vg.items())
                if ok and cfg == 5:
                    ok = fp.identify_copies({s + 1: v for s, v in vg.items()}) == j.chosen
                if not ok:                                  # say what failed (stderr; the line only carries the verdict)
                    bad = {s: (None if v is None or v[0] is None else "".join(map(str, v[0])), "".join(map(str, j.expected[s])))
                           for s, v in vg.items() if v is None or v[0] is None or not np.array_equal(v[0], j.expected[s])}
                    sys.stderr.write(f"votes_ok: step {g_} of a group of {size}: {len(votes)} votes, wrong (got, want): {bad}\n")
                    break
            return ok

    # steps per host iteration and graph capture: only shards too small to hide the host behind (module text)
    def group_policy(j):
        shard_bytes = j.n * H * W * 3
        small = 0 < shard_bytes < GRAPH_BELOW_BYTES
        if a.group:
            g_ = a.group
        elif small and j.equal:
            g_ = max(1, min(16, -(-300 * 1080 * 1920 * 3 // max(shard_bytes, 1))))
            while g_ > 1 and a.steps % g_:          # a divisor of the step count: every group of the timed region is a full one
                g_ -= 1
        else:
            g_ = 1
        if not j.equal:
            g_ = 1
        return g_, bool((small or a.graph) and not a.no_graph and a.streams == 1)

    G, use_graph = group_policy(job)
    runner = Runner(job, n_lanes=a.streams, group=G, graph=use_graph, emulate=bool(emu))
    lanes = runner.lanes
    n_chunks = runner.n_chunks
    chunk = runner.cf
    timed_steps = min(a.steps, 2000)                        # event pairs are pre-created; bound their number
    # event pairs on the DOMINANT kernel's launches only while `value` is timed (a pair on every launch of a step costs ~1.3 %
    # of the step: measured 267.5 k against 271.0 k frames/s, interleaved); the other kernels' durations come from a short pass
    # of their own after the timed region (`kernels`).  A graphed step carries no events (they ride on the dispatch, a graph
    # node has none): its dominant kernel is measured in that pass too.
    events_in_timed_region = not a.no_kernel_events and not use_graph
    # one pool for the warm-up AND the timed launches: nothing is collected (no host work, no extra idle time) between the warm-up
    # and the contract's barrier + synchronize; the warm-up's entries are dropped when the durations are read.  Idle gaps matter on
    # this device: after >= 5 ms of idleness the next launches run up to 25 % slower for ~10 ms, after ~1 ms a few per cent
    # (tools/idle_gap_experiment.py, profiles/r4_idle_gap.txt)
    warm_launches = n_chunks * a.warmup
    timing = _hip.Timing(n_chunks * (timed_steps + a.warmup) + 16, 1 << _hip.TIMING_KINDS.index(DOMINANT)) if events_in_timed_region else None
    opts_timed = timing.opts(flags) if timing else opts_plain

    runner.prepare()
    if grouped:                                     # first collective on the side stream: RCCL sets its channels up here
        with torch.cuda.stream(runner.side):
            p1 = lanes[0].pay[0, :1].reshape(-1, L)
            gather_payloads(p1.cpu() if a.backend == "gloo" else p1, equal_shards=job.equal, force=grouped)
        torch.cuda.synchronize()
    # the contract read literally -- W warm-up steps from idle, then K timed steps -- before the pre-heat, so that figure stays
    # available next to `value`: `value_no_preheat`.  Same fences, same collectives, every rank.
    cold = None
    if a.preheat_ms > 0 and not a.no_extras:
        if a.warmup:
            runner.run(a.warmup)
        el_c, v_c, sz_c = runner.timed(a.steps)
        cold = dict(el=el_c, votes_ok=runner.votes_ok(v_c, sz_c))
    preheat_ms = runner.preheat(a.preheat_ms)
    runner.set_opts(opts_timed)                             # every dominant-kernel launch of the timed steps carries its own event pair ...
    if a.warmup:
        runner.run(a.warmup)                                # ... and so do the warm-up steps: they are the timed steps' twins
    inject_failure("timed", rank)
    elapsed, votes, last_size = runner.timed(a.steps)
    own_elapsed = runner.own_elapsed
    shipped_order = lanes[0].eng.tile_order                  # what the timed region ran with
    shipped_info = lanes[0].eng.tile_order_info
    host_ms = {k: round(1e3 * v / a.steps, 4) for k, v in runner.host_s.items()}
    ranks_bound = int(bool(placement.get("bound")))
    if grouped:          # the slowest rank's host sets the pace: MAX over ranks (every rank takes part, unconditionally)
        hm = torch.tensor([host_ms["enqueue"], host_ms["vote"], float(ranks_bound)], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        hs = hm.clone()
        dist.all_reduce(hm, op=dist.ReduceOp.MAX)
        dist.all_reduce(hs, op=dist.ReduceOp.SUM)
        host_ms = dict(enqueue=round(float(hm[0]), 4), vote=round(float(hm[1]), 4), over="max over ranks")
        ranks_bound = int(round(float(hs[2])))
    runner.set_opts(opts_plain)
    kern, series = None, []
    if not a.no_kernel_events:
        kern = {}
        if timing:          # the dominant kernel's per-launch durations from the timed region itself (the warm-up's launches dropped)
            series = [m for m, k_ in timing.durations(n_chunks * (timed_steps + a.warmup) + 16) if k_ == DOMINANT][warm_launches:]
            timing.collect()
            timing.close()
            kern = {DOMINANT: dict(ms_total=float(np.sum(series)), launches=len(series))}
        # every kernel kind, from a short pass of its own straight after (plain launches, a pair on every one)
        kb = max(3, min(a.steps, 20))
        t_all = _hip.Timing(8 * n_chunks * kb + 16)
        runner.set_opts(t_all.opts(flags))
        runner.run(kb)
        torch.cuda.synchronize()
        for k_, v_ in t_all.collect().items():
            if k_ != DOMINANT or not kern.get(k_, {}).get("launches"):
                kern[k_] = v_
        runner.set_opts(opts_plain)
        t_all.close()

    # N > 1: what every rank measured by itself, so that one slow GPU shows in the one scaling run this project gets (the line's
    # ms_per_step is the MAX over ranks, which hides WHICH rank and by how much): one all-gather of three floats, every rank
    def _avg(kind):
        v_ = (kern or {}).get(kind) or {}
        return v_["ms_total"] / v_["launches"] if v_.get("launches") else float("nan")
    per_rank = None
    if grouped:
        mine = torch.tensor([1e3 * own_elapsed / a.steps, _avg(DOMINANT), _avg("analyze" if a.codec == "dct" and not planar else DOMINANT)],
                            dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        rows = np.stack([e_.cpu().numpy() for e_ in everyone])
        steps_ms = rows[:, 0]
        clean = lambda col: [None if np.isnan(x) else round(float(x), 5) for x in col]          # noqa: E731
        per_rank = dict(ms_per_step=clean(rows[:, 0]), dominant_kernel_ms=clean(rows[:, 1]), analyze_ms=clean(rows[:, 2]),
                        dominant_kernel=DOMINANT,
                        note="each rank's own figures, rank order: its K timed steps between the fences (host clock; the line's ms_per_step is the "
                             "MAX over ranks of the same span), its dominant kernel's mean launch duration in the timed region (HIP events), and "
                             "its analyze kernel's in the event pass straight after")
        extra_scaling = dict(slowest_rank=int(np.argmax(steps_ms)), slowest_ms_per_step=round(float(steps_ms.max()), 5),
                             median_ms_per_step=round(float(np.median(steps_ms)), 5), fastest_ms_per_step=round(float(steps_ms.min()), 5),
                             slowest_over_median=round(float(steps_ms.max() / np.median(steps_ms)), 4),
                             note="scaling efficiency is the driver's to compute; these say how much of a shortfall is ONE slow rank (placement of its "
                                  "frames in VRAM moves the kernels by 3-7 % on this device) rather than the path")

    # correctness of what was timed: every frame's payload, every segment's vote (and the leak's copy sequence)
    want_mine = job.expected_rows[job.first:job.first + n] if n else np.zeros((0, L), np.uint8)
    got_mine = runner.last_payloads().cpu().numpy() if n else np.zeros((0, L), np.uint8)
    ber = float((got_mine != want_mine).mean()) if n else 0.0
    votes_ok = runner.votes_ok(votes, last_size)
    payload_ok = bool((got_mine == want_mine).all())
    if cfg == 5:     # under the noise attack single frames may misread; what must hold is the vote -> copy sequence
        payload_ok = votes_ok

    extra = {}
    if per_rank is not None:
        extra["per_rank"] = per_rank
        extra["scaling_efficiency_inputs"] = extra_scaling
    if cold is not None:
        extra["value_no_preheat"] = round((n if emu else job.total_frames) * a.steps / cold["el"], 1)
        extra["no_preheat"] = dict(steps=a.steps, warmup=a.warmup, ms_per_step=round(1e3 * cold["el"] / a.steps, 4), votes_ok=cold["votes_ok"],
                                   note="the contract read literally: W warm-up steps from an idle device, then K timed steps, BEFORE the pre-heat; "
                                        "`value` is the same K steps after config.preheat_ms of untimed load (device at its operating state)")
    # a few frames of the timed batch and what the timed steps wrote for them, kept for the oracle check beside the CPU baseline
    # (the side measurements below reuse the output buffer).  Device-side copies only: a download here would idle the device for
    # milliseconds right in front of the second pass, and every idle gap of >= 5 ms is followed by ~15 ms of slower launches
    snap = None
    if a.codec == "dct" and mode == "embed_detect" and not planar and n and rank == 0 and not a.no_cpu_baseline and not emu:
        lane_last, _ = runner.last
        idx = sorted({0, n // 2, n - 1})
        sel = torch.as_tensor(idx, device=dev)
        snap = dict(idx=idx, frames=job.frames[sel], marked=lane_last.out[sel], rows=job.rows_dev[sel] if job.rows_dev is not None else None,
                    payloads=got_mine[idx])
    # Side measurements.  One GPU: each is guarded, a failure is reported inside the line.  N > 1: off unless --side-measurements
    # (every one of them is paid N-fold under barriers), and NEVER guarded: a rank that swallowed an exception would fall out of
    # step with the others' collectives and leave them in a barrier until the driver's time limit -- the exception propagates,
    # the rank exits non-zero, the launcher ends the job, no line is printed.
    sides_on = not a.no_extras and (world == 1 or a.side_measurements)

    class side:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            inject_failure(self.name, rank)

        def __exit__(self, et, ev, tb):
            if et is None or not issubclass(et, Exception) or world > 1:
                return False
            extra[self.name] = dict(error=repr(ev))
            return True

    if cfg == 5 and world == 1 and not a.no_extras and n and not emu:
        e0 = lanes[0].eng
        det = (lambda x: e0.detect(x, L, alpha=a.alpha)[0]) if a.codec == "dct" else (lambda x: e0.svd_detect(x, L, scale=15, blk=a.blk)[0])      # plain [n, L] counts
        extra["attacks"] = attack_suite(torch, det, job.leak_sample, job.keep, [job.expected[s] for s in range(S)], job.chosen, fp,
                                        vote_segments, deg, n_bits, H, W, "DCT" if a.codec == "dct" else f"DwtDctSvd(blk={a.blk})")
    # the same K steps once more, straight after the timed region.  `value` is the contract's figure (W warm-up steps after
    # idle, then K steps: with a short K that sits on the device's clock ramp); this one is the rate the device settles at.
    if not a.no_extras:
        inject_failure("second_pass", rank)
        el_b, v_b, sz_b = runner.timed(a.steps)
        extra["value_second_pass"] = round((n if emu else job.total_frames) * a.steps / el_b, 1)
        extra["second_pass"] = dict(steps=a.steps, ms_per_step=round(1e3 * el_b / a.steps, 4), votes_ok=runner.votes_ok(v_b, sz_b),
                                    note="the same K steps again straight after the timed region (no event pairs on the launches)")

    # ---- the fused mark kernel in BOTH tile orders, same K steps, interleaved in this process ----
    if a.codec == "dct" and mode == "embed_detect" and not planar and not a.separate_detect and sides_on and n and not a.no_kernel_events:
        try:
            with side("mark_order"):
                shipped, info = shipped_order, shipped_info
                t_ab = _hip.Timing(2 * n_chunks * a.steps + 16, 1 << _hip.TIMING_KINDS.index("mark_fused"))
                runner.set_opts(t_ab.opts(flags))
                res = {"xcd": [], "linear": []}
                for order in ("xcd", "linear", "linear", "xcd"):
                    runner.set_order(order)
                    runner.run(1)
                    torch.cuda.synchronize()
                    t_ab.collect()
                    el_o, _, _ = runner.timed(a.steps)
                    kk = t_ab.collect()["mark_fused"]
                    res[order].append((1e3 * el_o / a.steps, kk["ms_total"] / max(kk["launches"], 1)))
                t_ab.close()
                extra["mark_order"] = dict(
                    xcd_ms=round(float(np.mean([k for _, k in res["xcd"]])), 5), linear_ms=round(float(np.mean([k for _, k in res["linear"]])), 5),
                    xcd_step_ms=round(float(np.min([s_ for s_, _ in res["xcd"]])), 4), linear_step_ms=round(float(np.min([s_ for s_, _ in res["linear"]])), 4),
                    shipped=shipped, mode=a.tile_order, policy=info.get("policy"),
                    policy_detail={k: v for k, v in info.items() if k not in ("mode", "in_use", "policy")},
                    xcc_deal=engine_mod.probe_xcc_deal(dev),
                    note=f"fused mark kernel, average launch duration over 2 x {a.steps} steps per order, run xcd / linear / linear / xcd after the "
                         "timed region (*_step_ms: the faster of an order's two passes -- a pass now and then catches a one-off host stall of tens of "
                         "milliseconds); `shipped` is what the timed region used and `policy` how it was chosen (static rule = the library's rule on "
                         "the bytes per launch, no measurement; forced = --tile-order xcd / linear)")
        finally:
            runner.set_order(a.tile_order)
            runner.set_opts(opts_plain)

    # ---- emulation of an M-rank job: the whole job on this GPU, for the predicted speed-up --------------------------
    if emu:
        Gf, graph_f = group_policy(job_full)                  # the whole job as `bench.py --config C` runs it on one GPU
        full = Runner(job_full, n_lanes=1, group=Gf, graph=graph_f, emulate=False)
        full.prepare()
        full.run(max(1, a.warmup))
        el_f, v_f, sz_f = full.timed(a.steps)
        el_f2, _, _ = full.timed(a.steps)
        el_f = min(el_f, el_f2)
        full_ms = 1e3 * el_f / a.steps
        shard_ms = 1e3 * min(elapsed, el_b if not a.no_extras else elapsed) / a.steps
        speed = full_ms / shard_ms * (emu if job.scaling == "weak" else 1)
        extra["emulation"] = dict(
            world=emu, shard_frames=n, total_frames=job.total_frames, steps_per_host_iteration=G, hipgraph=bool(use_graph),
            shard_ms_per_step=round(shard_ms, 4), host_ms_per_step=host_ms, full_job_ms_per_step=round(full_ms, 4),
            full_job_votes_ok=full.votes_ok(v_f, sz_f), predicted_speedup=round(speed, 2),
            predicted_frames_per_s=round(job.total_frames / (shard_ms * 1e-3), 1),
            note=f"rank 0 of {emu}: its shard's embed/detect/payloads, a device copy into the pre-filled [{emu}, G, n, L] buffer where the RCCL "
                 f"all-gather would be, download, vote over all {job.total_frames} payloads per step; predicted speed-up = "
                 + ("M x " if job.scaling == "weak" else "") + "T(whole job on this GPU, as `--gpus 1` runs it) / T(rank 0's step), best of two passes each; a graphed "
                 "shard runs its steps on two graph branches, which a 300-frame single-stream step does not, so the ratio can exceed M.  Not modelled: "
                 "the real all-gather's latency (side stream, off the critical path) and N processes sharing the host")
        del full

    # the same steps alternating between TWO HIP streams (own workspace and output buffer each): independent batches overlap, one
    # step's analyze beside the other's mark+verify, launch gaps and kernel tails filled.  Reported next to `value`, which stays
    # single-stream: under concurrency a kernel's launch duration includes the time it shares the device, so the roofline object
    # (bytes per launch / launch duration) would no longer describe the kernel.
    if cfg in (2, 3) and a.codec == "dct" and a.streams == 1 and sides_on and n and not planar and not emu:
        try:
            with side("two_streams"):                            # may fail for want of room for the second output buffer
                runner.add_lane(torch.cuda.Stream())
                lanes[1].eng.workspace(H, W, lanes[1].eng._chunk(n, H, W))
                runner.run(2)
                el_t, v_t, sz_t = runner.timed(a.steps)
                extra["value_two_streams"] = round(job.total_frames * a.steps / el_t, 1)
                extra["two_streams"] = dict(steps=a.steps, ms_per_step=round(1e3 * el_t / a.steps, 4),
                                            path_frac_of_peak=round(job.total_frames * a.steps / el_t * 9 * H * W / 1e9 / (HBM_PEAK_GBPS * world), 4),
                                            votes_ok=runner.votes_ok(v_t, sz_t),
                                            note="the same K steps, consecutive steps on two HIP streams (python bench.py --streams 2 times this form)")
        finally:
            if len(lanes) > 1:
                torch.cuda.synchronize()
                lanes.pop()

    side_ok = cfg == 2 and a.codec == "dct" and not a.separate_detect and sides_on and not planar and not emu
    # second figure of the same line: SURVEY 8d config 2 read literally (embed, then the stand-alone detect)
    if side_ok:
        runner.set_opts(_hip.Opts(_hip.F_SEPARATE_DETECT, 0, None))
        k2 = max(3, min(a.steps, 20))
        runner.run(1)
        el2, v2, sz2 = runner.timed(k2)
        runner.set_opts(opts_plain)
        extra["value_separate_detect"] = round(world * n * k2 / el2, 1)
        extra["separate_detect"] = dict(steps=k2, ms_per_step=round(1e3 * el2 / k2, 4),
                                        note="embed, then the stand-alone detect on the written frames (12 B/px real traffic); "
                                             "bit-identical results", votes_ok=runner.votes_ok(v2, sz2))

    def side_rate(step, k):
        step()
        runner.fence()
        t0 = time.perf_counter()
        for _ in range(k):
            pm = step()
        runner.fence()
        return time.perf_counter() - t0, pm

    # the two operations the reference's drivers actually perform, each alone (VERDICT r4 missing 2): tests/mark.py:18-40 =
    # Embedder.__mark_frame per frame (video/embedder.py:33-39) -> embed only, 6 B/px algorithmic (the frame is read twice: the
    # luminance mask needs the frame mean first, so 9 B/px really move); tests/detect.py:17-31 = Extractor.__check_frame per frame
    # (video/extractor.py:30-34) -> detect + payloads, 3 B/px
    if cfg in (2, 3) and a.codec == "dct" and mode == "embed_detect" and not planar and sides_on and n and not emu:
        with side("embed_only"):
            e0, k5 = lanes[0].eng, max(3, min(a.steps, 20))
            marked = lanes[0].out
            pay5 = torch.empty((n, L), dtype=torch.uint8, device=dev)

            def embed_step():
                return e0.embed(job.frames, job.wm_dev, alpha=a.alpha, wm_row=job.rows_dev, out=marked)

            def detect_step():
                return e0.payloads(e0.detect(marked, L, alpha=a.alpha)[0], N, perm_dev, out=pay5)
            # the kernel tests/mark.py's operation runs -- mark_rgb8_kernel WITHOUT the fused verify -- gets its own event pairs here,
            # so `kernels.mark` (duration, achieved GB/s, fraction of peak) is in every default line (VERDICT r5 missing 2)
            t_mark = _hip.Timing(n_chunks * (k5 + 2) + 16, 1 << _hip.TIMING_KINDS.index("mark")) if kern is not None else None
            if t_mark:
                e0.opts = t_mark.opts(flags)
            for key, step, bpp_alg, what in (("embed_only", embed_step, 6, "embed alone: analyze + mark (no verify), marked frames written"),
                                             ("detect_only", detect_step, 3, "detect + payloads of the marked frames alone: analyze, finalize, payload kernel")):
                el5, res5 = side_rate(step, k5)
                rate = world * n * k5 / el5
                extra[key] = dict(value=round(rate, 1), unit="frames/s", steps=k5, ms_per_step=round(1e3 * el5 / k5, 4),
                                  algorithmic_bytes_per_frame=bpp_alg * H * W, algorithmic_GBps=round(rate * bpp_alg * H * W / 1e9, 1),
                                  frac_of_peak=round(rate * bpp_alg * H * W / 1e9 / (HBM_PEAK_GBPS * world), 4), note=what)
                if key == "detect_only":
                    extra[key]["payload_ok"] = bool((res5.cpu().numpy() == want_mine).all())
                if key == "embed_only" and t_mark:
                    e0.opts = opts_plain
                    got_mark = t_mark.collect().get("mark")
                    if got_mark and got_mark["launches"]:
                        kern["mark"] = got_mark
                    t_mark.close()
            del pay5

    # planar 4:2:0 frames through the same step, HBM-resident: what the fused ingest/egress costs or saves
    if side_ok and H % 8 == 0 and W % 8 == 0:
        e0 = lanes[0].eng
        planes = e0.rgb_to_yuv420(job.frames)
        pout = torch.empty_like(planes)

        def planar_step():
            _, c, _ = e0.embed_detect_yuv420(planes, H, W, job.wm_dev, L, alpha=a.alpha, out=pout)
            return e0.payloads(c, N, perm_dev)
        k3 = max(3, min(a.steps, 20))
        el3, pm = side_rate(planar_step, k3)
        extra["planar_i420"] = dict(value=round(world * n * k3 / el3, 1), unit="frames/s", steps=k3, ms_per_step=round(1e3 * el3 / k3, 4),
                                    payload_ok=bool((pm.cpu().numpy() == PAYLOAD[None]).all()),
                                    note="embed+detect on I420 planes (1.5 B/px in, 1.5 B/px out, conversion fused into the kernels); "
                                         "payload read from the WRITTEN planes, i.e. after 4:2:0 subsampling")
        del planes, pout

    # the codec tests/mark.py and tests/detect.py construct, same frames, same step: embed + verify + payloads
    if side_ok:
        e0 = lanes[0].eng
        k4 = max(3, min(a.steps, 20))
        for blk, key in ((4, "dwtdctsvd"), (8, "dwtdctsvd_blk8")):
            def svd_step(blk=blk):
                _, c, _ = e0.svd_embed_detect(job.frames, job.wm_dev, L=L, scale=15, wm_row=job.rows_dev, out=lanes[0].out, blk=blk, partial=True)
                return e0.payloads(c, DctEngine.svd_bits_per_frame(H, W, blk), perm_dev)
            el4, pm = side_rate(svd_step, k4)
            if snap is not None:                       # the same three frames as the DCT check, as this codec marked them (device-side copy)
                snap[key] = (lanes[0].out[torch.as_tensor(snap["idx"], device=dev)], pm[snap["idx"]].cpu().numpy(), blk)
            extra[key] = dict(value=round(world * n * k4 / el4, 1), unit="frames/s", steps=k4, ms_per_step=round(1e3 * el4 / k4, 4),
                              payload_ok=bool((pm.cpu().numpy() == PAYLOAD[None]).all()),
                              algorithmic_GBps=round(n * k4 * 6 * H * W / el4 / 1e9, 1),
                              note="DwtDctSvd embed + verify + payloads in one 6 B/px pass (scale 15)" if blk == 4 else
                                   "DwtDctSvd(blk=8) embed + verify + payloads (16x16 pixel tiles, an 8x8 singular-triplet solve per tile; H*W/256 bits per frame)")

    if rank != 0:
        if world > 1:
            barrier()
            dist.destroy_process_group()
        return

    # what the device is doing under this load: one rocm-smi sample (engine clock, socket power) while the steps keep running --
    # the evidence behind "the socket sits on its power limit" travels with the line
    if side_ok and world == 1:
        try:
            import re
            import shutil
            import threading
            smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
            stop = threading.Event()

            def keep_busy():
                torch.cuda.set_device(dev)
                while not stop.is_set():
                    for _ in range(50):
                        runner.hot_path(lanes[0])
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
    if side_ok and world == 1 and (H, W) == (1080, 1920):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import pcie_pipeline
            pc = {}
            for fmt in ("rgb24", "i420"):
                f_, g_ = pcie_pipeline.measure(fmt, n=200, B=50, H=H, W=W, eng=lanes[0].eng)
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
    # the literal per-frame plugin boundary on host float32 YUV frames
    if side_ok and world == 1:
        try:
            extra["plugin_yuv32f"] = plugin_yuv32f_rates(H, W, a.alpha, job.frames[0].cpu().numpy())
        except Exception as exc:
            extra["plugin_yuv32f"] = dict(error=repr(exc))

    # achievable HBM bandwidth of this device, same run: 16-byte streaming copy (read + write) and read-only stream
    out0 = lanes[0].out
    if job.frames.numel() >= (1 << 28) and out0 is not None and out0.numel() == job.frames.numel():
        probe_src, probe_dst = job.frames, out0
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

    for key, ceil_name, ceil in (("embed_only", "frac_of_measured_copy", copy_gbps), ("detect_only", "frac_of_measured_read", read_gbps)):
        if key in extra and "algorithmic_GBps" in extra[key]:
            extra[key][ceil_name] = round(extra[key]["algorithmic_GBps"] / (ceil * world), 4)
    units = n if emu else job.total_frames                  # emulation: `value` is what this ONE GPU really processed
    fps = units * a.steps / elapsed
    frame_bytes = 3 * H * W
    roof = None
    sha = source_sha16()
    if kern:
        # algorithmic bytes per frame and kernel: analyze reads the frame (3 B/px);
        # mark reads it again and writes the marked frame (6 B/px); the fused mark+verify kernel
        # moves the same 6 B/px and spares detect's 3 B/px read.  Sum over a step = 9 B/px.  Planar 4:2:0: half of each.
        svd_bytes = frame_bytes if mode == "detect" else 2 * frame_bytes
        alg = {"analyze": frame_bytes, "mark": 2 * frame_bytes, "mark_fused": 2 * frame_bytes, "svd": svd_bytes,
               "planar_analyze": frame_bytes // 2, "planar_mark": frame_bytes}
        svd_name = ("svd8_rgb8_kernel" if a.blk == 8 else "svd_rgb8_kernel") + ("<detect>" if mode == "detect" else "<embed+verify>")
        names = {"analyze": "analyze_kernel<rgb8>", "mark": "mark_rgb8_kernel", "mark_fused": "mark_rgb8_kernel<fused verify>",
                 "svd": svd_name, "planar_analyze": f"analyze_yuv420_kernel<{a.pixfmt}>", "planar_mark": f"mark_yuv420_kernel<{a.pixfmt}, fused verify>"}
        ceiling = {"analyze": read_gbps, "planar_analyze": read_gbps}
        per = {}
        for k, v in kern.items():
            if not v["launches"]:
                continue
            avg_ms = v["ms_total"] / v["launches"]
            # one launch of each DCT / planar kind per chunk and step (the DwtDctSvd codec needs no workspace and does not chunk: one launch per
            # step); the event pool is bounded, so a very long run records its first launches only
            per_step = 1 if k == "svd" else n_chunks
            d = dict(avg_launch_ms=round(avg_ms, 5), launches=v["launches"], ms_per_step=round(avg_ms * per_step, 4))
            if k in alg:
                frames_per_launch = n / per_step
                d["algorithmic_bytes_per_launch"] = int(frames_per_launch * alg[k])
                d["achieved_GBps"] = round(frames_per_launch * alg[k] / (avg_ms * 1e-3) / 1e9, 1)
                d["frac_of_peak"] = round(d["achieved_GBps"] / HBM_PEAK_GBPS, 4)
                d["frac_of_measured_" + ("read" if k in ceiling else "copy")] = round(d["achieved_GBps"] / ceiling.get(k, copy_gbps), 4)
            per[k] = d
        extra["kernels"] = per
        extra["kernels_note"] = ((f"{DOMINANT}: event pairs on its launches in the timed region; the other kinds: " if timing else
                                  "the timed region replays a captured hipGraph (no events on graph nodes); every kind: ")
                                 + f"a pass of {max(3, min(a.steps, 20))} steps straight after it with a pair on every launch")
        extra["kernel_ms_per_step"] = round(sum(v["ms_per_step"] for v in per.values()), 4)
        if per:
            dom = DOMINANT if DOMINANT in per else max((k for k in per if k in alg), key=lambda k: per[k]["ms_per_step"])
            achieved = per[dom]["achieved_GBps"]
            # PMC-measured HBM bytes per launch (separate rocprofv3 passes, tools/prof.sh): quoted only when that profile
            # was taken from exactly these kernel sources and this frame size
            traffic, traffic_src = None, None
            for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
                tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                if tj.get("source_sha16") == sha and (tj["height"], tj["width"]) == (H, W) and dom in tj \
                        and names[dom].split("<")[0] in tj.get("kernel_names", {}).get(dom, names[dom]):
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
            if series:          # how the dominant kernel's duration moves through the timed region, launch by launch
                q = max(1, len(series) // 4)
                roof["launch_ms_through_the_timed_region"] = dict(
                    first_quarter=round(float(np.mean(series[:q])), 5), last_quarter=round(float(np.mean(series[-q:])), 5),
                    min=round(float(np.min(series)), 5), median=round(float(np.median(series)), 5), max=round(float(np.max(series)), 5),
                    first_launches=[round(x, 4) for x in series[:6]])

    # 9 B/px per embed+detect frame with the DCT codec (4.5 on 4:2:0 planes); the DwtDctSvd codec has no frame-global
    # dependency and no separate detect read: 6 B/px; detect only: 3 B/px
    bpp = 3 if mode == "detect" else (4.5 if planar else 9) if a.codec == "dct" else 6
    path_gbps = fps * bpp * H * W / 1e9
    what = {2: "configs[1]", 3: "configs[2]", 4: "configs[3]", 5: "configs[4]"}[cfg]
    op = "embed+detect" if mode == "embed_detect" else "leak detect"
    codec_name = "DCT" if a.codec == "dct" else (f"DwtDctSvd(blk={a.blk})" if a.blk != 4 else "DwtDctSvd")
    sw = job.shard_world
    if cfg in (2, 3):
        workload = f"synthetic {W}x{H} u8 {'RGB' if not planar else a.pixfmt.upper() + ' planes'} x{n} frames per GPU, "
    else:
        workload = (f"synthetic {W}x{H} u8 RGB, {S} segments x {F} frames sharded over {sw} rank(s), "
                    + ("own payload per segment, " if cfg == 4 else f"{C} copies per segment, leak 01201201 + N(0,2) noise, "))
    if emu:
        workload = f"RANK 0 OF AN EMULATED {emu}-RANK JOB on one GPU: " + workload

    line = {
        "metric": f"1080p frames/sec {op}" if (H, W) == (1080, 1920) else f"{W}x{H} frames/sec {op}",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": job.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload + f"{codec_name} {op}+vote (BASELINE.json {what})",
                   "codec": a.codec, "frames_per_gpu": n, "payload_bits": L, "alpha": a.alpha,
                   "chunk_frames": chunk, "chunks_per_step": n_chunks, "steps_per_host_iteration": G, "hipgraph": bool(use_graph),
                   "preheat_ms": preheat_ms, "placement_probe": runner.placement,
                   "tile_order": shipped_order if (a.codec == "dct" and mode == "embed_detect" and not planar) else None,
                   "tile_order_policy": shipped_info.get("policy") if (a.codec == "dct" and mode == "embed_detect" and not planar) else None,
                   "detect": ("stand-alone kernels" if (a.separate_detect or mode == "detect") else "fused into the mark kernel")
                   if a.codec == "dct" else ("stand-alone kernel" if mode == "detect" else "fused into the embed kernel"),
                   "sharding": f"{'frames' if cfg in (2, 3) else 'segments'}, {sw} rank(s), one RCCL all-gather of payloads"},
        "payload_ber": ber, "payload_bit_exact": payload_ok and votes_ok,
        "roofline": roof,
        "path": {"algorithmic_GBps": round(path_gbps, 1), "bytes_per_frame": bpp * H * W,
                 "frac_of_peak": round(path_gbps / (HBM_PEAK_GBPS * world), 4),
                 "frac_of_measured_copy": round(path_gbps / (copy_gbps * world), 4)},
        "hbm_copy_GBps": round(copy_gbps, 1), "hbm_read_GBps": round(read_gbps, 1),
        "source_sha16": sha,
        "collective": {"backend": ("rccl" if a.backend == "nccl" else a.backend) if grouped else None, "ranks": ranks_seen,
                       "self_launched": bool(os.environ.get("OFMK_BENCH_SELF_LAUNCHED")), "env": rank_env},
        "rccl_ranks": ranks_seen if (grouped and a.backend == "nccl") else None,
        "host_ms_per_step": host_ms,            # CPU time issuing a step / voting on one (N > 1: the slowest rank's); must stay < ms_per_step
        "placement": dict(placement, ranks_bound=ranks_bound),
        "cpu_baseline": None,
    }
    line.update(extra)

    # every rank is past its timed region and its side measurements: let the others go BEFORE the CPU baseline occupies this
    # host for tens of seconds (so N > 1 lines carry it too), then print the one line
    if grouped:
        barrier()
        dist.destroy_process_group()
    if not a.no_cpu_baseline and a.codec == "dct" and mode == "embed_detect":
        try:
            if affinity_at_start:                      # the CPU baseline may use every core this process was given, not just the GPU-local ones
                os.sched_setaffinity(0, affinity_at_start)
            nb = 96 if H * W <= 1920 * 1080 else 24
            wm_cpu = Shuffler(key=0).generate_wm(PAYLOAD, (1, N))
            line["cpu_baseline"] = cpu_baseline(synthetic_frames(nb, H, W, seed=2000, device=dev).cpu().numpy(), wm_cpu, a.alpha, a.cpu_seconds)
        except Exception as exc:                       # e.g. no C compiler on the box: report, do not lose the GPU line
            line["cpu_baseline"] = dict(value=None, unit="frames/s", cores=0, kind="port", sample=f"cpu baseline failed: {exc!r}")
        if snap is not None:                           # the timed workload's own frames against the oracle (beside the baseline: the C oracle is loaded)
            try:
                e0 = lanes[0].eng

                def gpu_bits(ref_marked):
                    return e0.detect(torch.from_numpy(ref_marked).to(dev), L, alpha=a.alpha, want_bits=True)[1].cpu().numpy()
                rows_sel = snap["rows"].cpu().numpy() if snap["rows"] is not None else np.zeros(len(snap["idx"]), np.int64)
                line["oracle_check"] = oracle_check(snap["frames"].cpu().numpy(), snap["marked"].cpu().numpy(), gpu_bits,
                                                    [job.wm_table[int(r)] for r in rows_sel], a.alpha, snap["payloads"])
                line["oracle_check"]["frame_indices"] = snap["idx"]
                if not line["oracle_check"]["within_budget"]:
                    line["payload_bit_exact"] = False
                for key in ("dwtdctsvd", "dwtdctsvd_blk8"):      # the codec mark.py / detect.py construct: the same frames against the NumPy oracle
                    if key in snap and key in line:
                        marked_svd, pay_svd, blk_ = snap[key]

                        def gpu_bits_svd(ref_marked, blk_=blk_):
                            return e0.svd_detect(torch.from_numpy(ref_marked).to(dev), L, scale=15, blk=blk_, want_bits=True)[1].cpu().numpy()
                        line[key]["oracle_check"] = oracle_check_svd(snap["frames"].cpu().numpy(), marked_svd.cpu().numpy(), gpu_bits_svd,
                                                                     [job.wm_table[int(r)] for r in rows_sel], pay_svd, blk_)
            except Exception as exc:
                line["oracle_check"] = dict(error=repr(exc))
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    try:
        main()
    except Exception:
        # N > 1: this rank must not linger (its peers are inside collectives that will never complete): report, leave at once with a
        # failure code, and let the launcher end the others -- no interpreter shutdown that could wait on a communicator
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        if int(os.environ.get("WORLD_SIZE", 1)) > 1:
            os._exit(1)
        raise SystemExit(1)
