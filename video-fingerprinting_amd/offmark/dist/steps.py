"""Issuing many steps of one batch shape: lanes, grouped hipGraph replay, side-stream gather + download, host vote.

The reference marks and reads one frame at a time (src/offmark/video/embedder.py:18-31, extractor.py:18-28) and votes on a
segment when it is done (tests/segment_mark_detect_hls.py:126-155).  On a GPU a SEGMENT is a small unit of work -- 48 frames
of 1080p are 0.2 ms of embed + detect, 0.07 ms of detect -- less than a host needs to issue one step (kernel launches,
an all-gather, a download, a vote: ~0.15 ms).  When segments shard over the GPUs of a node (one per rank, BASELINE.json
configs[3] / [4]) the host, not the device, would set the rate.  `StepPipeline` takes the host out of such steps:

  * G steps per host iteration: the G steps' kernels are replayed as ONE captured hipGraph (the C ABI only enqueues kernels, so
    whole call sequences capture: include/offmark_hip.h), their payloads land in G slots of one device buffer, ONE
    all-gather (RCCL over xGMI when the process group is "nccl") and ONE download move them, and the host votes on all G
    steps' rows in one vectorised call (offmark.dist.vote.group_segment_ids);
  * inside the graph the steps alternate between TWO branches (own engine = own workspace, own output buffer), so that one
    step's tail and inter-kernel gaps hide under the other's kernels (worth 8-12 % at 48 frames per step);
  * payload slots are double-buffered (one graph per half): a group's payloads stay untouched while the side stream gathers and
    downloads them and the lane already replays the next group;
  * the gather, the download and the vote of group g overlap the kernels of group g + 1.

With G = 1 and no graph this is the plain double-buffered step loop bench.py has always run for large batches.  The caller
supplies `issue(engine, out, payload_slot)`, which enqueues ONE step on the current stream and leaves its [n, L] uint8
payloads in `payload_slot`; everything else (what a step is) stays with the caller.
(Round 5 tried the payload epilogue once per group on the side stream instead of as every step's last launch: no gain at 300
frames per step, and detect-only steps lost 4-6 % to the concurrent launch -- profiles/r5_frame_tail_experiment.txt.)
"""
from __future__ import annotations

import time
from types import SimpleNamespace

import numpy as np

from .vote import gather_payloads, group_segment_ids, vote_segments


class StepPipeline:
    def __init__(self, device, n: int, L: int, segment_ids, make_engine, make_out, issue, lanes: int = 1, group: int = 1,
                 graph: bool = False, gather=None, equal_shards: bool = True, force_collective: bool = False):
        """n: frames (payload rows) this rank contributes per step; segment_ids: the segment of every row of ONE step's
        gathered result, rank-major ([ranks * n]); make_engine() / make_out(): a fresh engine / output buffer (a second pair is
        made for the graph's second branch); issue(engine, out, slot): enqueue one step; gather(rows [size * n, L], g, size) ->
        the gathered rows of group g (default: offmark.dist.vote.gather_payloads)."""
        self.n, self.L, self.G, self.use_graph = int(n), int(L), max(1, int(group)), bool(graph)
        self.seg = np.asarray(segment_ids)
        self.total = int(self.seg.size)
        # Grouped steps lay the gathered rows out as [rank][step][n] with the SAME n on every rank (group_segment_ids): ragged
        # shards (or a rank with nothing to do) would put segment ids on the wrong rows and mix steps in the vote, silently.
        if self.G > 1 and (not equal_shards or self.n <= 0 or self.total % self.n != 0):
            raise ValueError(f"group={self.G} needs equal, non-empty shards on every rank (n={self.n}, gathered rows per step={self.total}, "
                             f"equal_shards={equal_shards}): use group=1 for ragged shards")
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.ranks = self.total // max(self.n, 1) if self.n else 1
        self.S_ids = int(self.seg.max()) + 1 if self.total else 1
        self.make_engine, self.make_out, self.issue = make_engine, make_out, issue
        self.gather = gather or (lambda rows, g, size: gather_payloads(rows, equal_shards=equal_shards, force=force_collective))
        self.lanes = []
        for i in range(lanes):
            self.add_lane(torch.cuda.current_stream() if i == 0 else torch.cuda.Stream())
        self.side = torch.cuda.Stream()                # all-gather + download: off the compute stream, so a slow peer never
        self.handoff = [torch.cuda.Event() for _ in range(2)]         # stalls this rank's next step
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.host = [torch.empty((self.total * self.G, self.L), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.host_s = {"enqueue": 0.0, "vote": 0.0}    # host-side seconds spent issuing work / voting (not waiting)
        self._ids = {}
        self.last = None

    # -- lanes -------------------------------------------------------------------------------------
    def add_lane(self, stream):
        t = self.torch
        two = self.use_graph and self.G >= 2 and self.n > 0       # the graph's second branch needs its own workspace and output
        lane = SimpleNamespace(eng=self.make_engine(), out=self.make_out(), eng2=self.make_engine() if two else None,
                               out2=self.make_out() if two else None, stream=stream, graph=[None, None],
                               pay=t.empty((2, self.G, max(self.n, 0), self.L), dtype=t.uint8, device=self.device))
        self.lanes.append(lane)
        return lane

    def engines(self):
        return [e for lane in self.lanes for e in (lane.eng, lane.eng2) if e is not None]

    def drop_graphs(self):
        for lane in self.lanes:
            lane.graph = [None, None]

    def step(self, lane, slot=0, par=0, branch=0):
        """One step on the current stream -> lane.pay[par, slot]."""
        if self.n == 0:
            return
        e, out = (lane.eng, lane.out) if branch == 0 else (lane.eng2, lane.out2)
        self.issue(e, out, lane.pay[par, slot])

    # -- set-up --------------------------------------------------------------------------------------
    def warm_download_path(self):
        """First use of the download path (side stream, page-locked landing buffers, events) BEFORE the device is brought to its
        operating state: in a kernel trace the first non-blocking D2H copy of a process held the host for milliseconds (6 ms under
        rocprofv3) right after the first step, and on MI355X every idle gap of >= 5 ms is followed by ~15 ms of slower launches
        (profiles/r4_idle_gap.txt) -- exactly where a short measurement sits."""
        t = self.torch
        with t.cuda.stream(self.side):
            scratch = t.zeros((1, self.L), dtype=t.uint8, device=self.device)
            for h, ev in zip(self.host, self.ready):
                if h.shape[0]:
                    h[:1].copy_(scratch, non_blocking=True)
                ev.record()
            for ev in self.handoff:
                ev.record()
        t.cuda.synchronize()

    def prepare(self):
        """One full step on every engine (allocations, code objects), then the graphs."""
        t = self.torch
        self.warm_download_path()
        for lane in self.lanes:
            for b, e in enumerate((lane.eng, lane.eng2)):
                if e is not None:
                    with t.cuda.stream(lane.stream):
                        self.step(lane, branch=b)
        t.cuda.synchronize()
        if self.use_graph and self.n:
            for lane in self.lanes:
                for par in (0, 1):
                    self.capture(lane, par)

    def capture(self, lane, par):
        """G steps of this lane as ONE hipGraph, odd steps on a forked second branch."""
        t = self.torch
        cs, fork = t.cuda.Stream(), t.cuda.Stream()
        two = lane.eng2 is not None

        def body():
            if two:
                fork.wait_stream(cs)                                # fork
            for g_ in range(self.G):
                if two and g_ % 2:
                    with t.cuda.stream(fork):
                        self.step(lane, g_, par, branch=1)
                else:
                    self.step(lane, g_, par)
            if two:
                cs.wait_stream(fork)                                # join
        with t.cuda.stream(cs):
            body()
            t.cuda.synchronize()
            graph = t.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are checked against the capture -- a collective library's watchdog thread
            # (RCCL: event queries on its own streams) must not invalidate it
            with t.cuda.graph(graph, stream=cs, capture_error_mode="thread_local"):
                body()
        t.cuda.synchronize()
        lane.graph[par] = graph

    # -- the loop ------------------------------------------------------------------------------------
    def ids_for(self, size):
        if size not in self._ids:
            self._ids[size] = group_segment_ids(self.seg, self.ranks, size)
        return self._ids[size]

    def graph_allowed(self, lane):
        """A launch that carries HIP events (ofmk_timing) cannot be a graph node: such steps are issued as plain launches."""
        o = getattr(lane.eng, "opts", None)
        return not (o is not None and o.timing)

    def enqueue(self, g, size):
        """GPU half of group g (`size` steps); then, on the side stream, the gather of the payloads and their download into
        page-locked memory.  Returns the number of gathered rows."""
        t = self.torch
        t_in = time.perf_counter()
        lane = self.lanes[g % len(self.lanes)]
        par = (g // len(self.lanes)) & 1
        with t.cuda.stream(lane.stream):
            if self.use_graph and size == self.G and self.n and self.graph_allowed(lane):
                if lane.graph[par] is None:
                    self.capture(lane, par)
                lane.graph[par].replay()
            else:
                for g_ in range(size):
                    self.step(lane, g_, par)
            self.handoff[g & 1].record()
        self.last = (lane, par)
        with t.cuda.stream(self.side):
            self.side.wait_event(self.handoff[g & 1])
            everyone = self.gather(lane.pay[par, :size].reshape(size * self.n, self.L), g, size)
            self.host[g & 1][:everyone.shape[0]].copy_(everyone, non_blocking=True)
            self.ready[g & 1].record()
        self.host_s["enqueue"] += time.perf_counter() - t_in
        return everyone.shape[0]

    def finish(self, g, size, rows):
        """Host half of group g: the reference's cross-frame Counter vote for every (step, segment) of the group, once its payloads
        have landed.  Runs while the GPU is already working on group g + 1.  Keys: segment s of step k votes under k * S + s."""
        self.ready[g & 1].synchronize()
        t_in = time.perf_counter()
        v = vote_segments(self.host[g & 1][:rows].numpy(), self.ids_for(size))
        self.host_s["vote"] += time.perf_counter() - t_in
        return v

    def plan(self, steps):
        full, rest = divmod(steps, self.G)
        return [self.G] * full + ([rest] if rest else [])

    def run(self, steps):
        """`steps` steps in groups; returns (the last group's votes, that group's size)."""
        prev = None
        for g, size in enumerate(self.plan(steps)):
            rows = self.enqueue(g, size)
            if prev is not None:
                self.finish(*prev)
            prev = (g, size, rows)
        return self.finish(*prev), prev[1]

    def last_payloads(self):
        """Device payloads [n, L] of the first step of the group issued last."""
        lane, par = self.last
        return lane.pay[par, 0]
