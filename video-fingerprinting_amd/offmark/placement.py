"""Placement-probed buffers: the engine's workspace and the marked frames' destination, picked among a few candidate allocations.

On MI355X the frame kernels run at different speeds depending on WHERE THE BUFFERS THEY WRITE live: the same launch over the same
input takes analyze 0.347 / 0.359 / 0.371-0.387 ms with its records in different workspace allocations of one process, and the
fused mark kernel 0.663 / 0.677 / 0.688 ms into different output allocations (profiles/r6_placement_candidates.txt; the analyze
ladder of profiles/r6_mark_ladder.txt shows that the whole fast / slow difference between processes sits in the record WRITES: with
the records kept inside L2 every box runs analyze at 0.285 ms).  A plain streaming read or copy does not see these levels, the
driver offers no way to ask for one, and a freed and re-allocated buffer can come back on another -- but a buffer keeps its level
while it lives, the levels can be MEASURED in milliseconds, and both buffers are the engine's to allocate.  So: K candidates alive
at once (torch hands out cached blocks first, then fresh ones: different physical places), the real kernel over the caller's own
frames into each, keep the fastest, drop the rest.

The reference has no counterpart (host NumPy arrays, one frame at a time: src/offmark/video/embedder.py:18-31).  Set-up of the
device-resident batch path only; no result depends on it (the chosen buffers hold what any others would).
"""
from __future__ import annotations

import time

import numpy as np

from . import _hip

DEFAULT_CANDIDATES = 8
MIN_BYTES = 256 << 20         # batches smaller than this are launch-bound: nothing to gain


def _pick(torch, candidates, run, rounds=3, launches=3, preheat_ms=60.0):
    """Median launch time of run(candidate) per candidate, rounds interleaved in alternating direction after a short pre-heat
    (a device coming out of idle speeds up for ~100 ms: without it the candidates probed last would win).  -> (best index, ms[])"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    while 1e3 * (time.perf_counter() - t0) < preheat_ms:
        for _ in range(8):
            run(candidates[0])
        torch.cuda.current_stream().synchronize()
    ms = np.zeros((rounds, len(candidates)))
    for r in range(rounds):
        order = range(len(candidates)) if r % 2 == 0 else range(len(candidates) - 1, -1, -1)
        for i in order:
            run(candidates[i])                                   # untimed first touch
            e0.record()
            for _ in range(launches):
                run(candidates[i])
            e1.record()
            e1.synchronize()
            ms[r, i] = e0.elapsed_time(e1) / launches
    med = np.median(ms, axis=0)
    return int(np.argmin(med)), med


def place_buffers(engine, frames, want_out: bool = True, candidates: int = DEFAULT_CANDIDATES, max_bytes: int | None = None):
    """frames: contiguous CUDA uint8 [n, H, W, 3] -- the batch (or one like it, in the same allocation) the engine will work on.
    Picks the engine's workspace for this frame size among `candidates` allocations by the analyze kernel's time over `frames`,
    then (want_out) the destination of the marked frames among `candidates` allocations by the fused mark + verify kernel's time;
    returns (out tensor or None, report).  Synchronises; call once at set-up, before capturing graphs.  `max_bytes`: upper bound
    on what the output candidates may take together (default: a third of the device's free memory)."""
    t = engine.torch
    n, H, W = engine._check_frames(frames, t.uint8)
    report = dict(candidates=int(candidates))
    out = None
    if candidates < 2 or frames.numel() < MIN_BYTES:
        report["note"] = "off" if candidates < 2 else "batch too small to matter"
        return (t.empty_like(frames) if want_out else None), report
    lib, s = engine.lib, _hip.current_stream()
    cf = engine._chunk(n, H, W)
    part = frames[:cf]
    nbytes = lib.ofmk_workspace_bytes(cf, H, W)
    with t.cuda.device(engine.device):
        wss = [t.empty(nbytes, dtype=t.uint8, device=engine.device) for _ in range(candidates)]

        def analyze(ws):
            _hip.check(lib.ofmk_stage_analyze_rgb8(part.data_ptr(), cf, H, W, ws.data_ptr(), ws.numel(), s, None))
        best, ms = _pick(t, wss, analyze)
        ws = wss[best]
        engine._ws = {(H, W): ws}                                 # what engine.workspace(H, W, <= cf) hands out from now on
        report["workspace"] = dict(chosen=best, analyze_ms=[round(float(x), 4) for x in ms])
        del wss
        if want_out:
            free, _ = t.cuda.mem_get_info(engine.device)
            budget = free // 3 if max_bytes is None else int(max_bytes)
            k = int(max(1, min(candidates, budget // max(frames.numel(), 1))))
            outs = [t.empty_like(frames) for _ in range(k)]
            wm = t.zeros((1, H * W // 64), dtype=t.uint8, device=engine.device)
            wm[0, ::2] = 1
            analyze(ws)                                          # the records the mark kernel reads

            def mark(o):
                _hip.check(lib.ofmk_stage_mark_rgb8(part.data_ptr(), o.data_ptr(), cf, H, W, wm.data_ptr(), 20.0, 1, ws.data_ptr(), ws.numel(), s, None))
            if k > 1:
                best_o, ms_o = _pick(t, outs, mark, preheat_ms=0.0)
            else:
                best_o, ms_o = 0, np.zeros(1)
            out = outs[best_o]
            report["output"] = dict(chosen=best_o, fused_mark_ms=[round(float(x), 4) for x in ms_o])
            del outs
    t.cuda.current_stream().synchronize()
    report["note"] = ("engine-owned buffers picked among candidate allocations by the real kernels' launch time over the caller's frames "
                      "(median of 3 interleaved rounds of 3 launches); the losers go back to the allocator")
    return out, report
