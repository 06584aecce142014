"""Frame/segment sharding across GPUs and the cross-frame payload vote.

The reference has no parallelism (video/embedder.py:19-27 is one frame at a time); frames are
independent, so each rank marks/reads its own contiguous shard and no frame ever crosses xGMI.
The only exchange is the reference's cross-frame step: the per-frame recovered payloads are
collected and the most common whole pattern wins (PatternCollectorExtractor,
tests/segment_mark_detect_hls.py:126-155).  Here: one all-gather of [n_local, L] uint8 (RCCL when
the backend is "nccl", gloo on CPU for tests), then the same Counter vote on every rank.
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) of ``rank``; the first n_total % world ranks get one extra item."""
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def init_from_env(backend: str | None = None, force: bool = False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
    Returns (rank, world).  With WORLD_SIZE unset or 1 no group is created unless ``force`` (a one-rank
    group on 127.0.0.1, for dry runs of the collective path)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if force and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def payloads_from_counts(counts, n_bits: int, perm):
    """Device-side DeShuffler epilogue (de_shuffler.py:17-22) for a batch: counts [n, L] (ones among
    bits[i::L]) -> uint8 [n, L].  float64 like the reference; stays on ``counts``' device."""
    import torch
    L = counts.shape[-1]
    i = torch.arange(L, device=counts.device)
    lens = torch.where(i < n_bits, (n_bits - i + L - 1) // L, torch.zeros_like(i)).to(torch.float64)
    means = counts.to(torch.float64) / lens
    perm = torch.as_tensor(np.asarray(perm), device=counts.device, dtype=torch.long)
    payload = torch.empty_like(means)
    payload[..., perm] = means
    thr = 0.5 * (payload.max(dim=-1, keepdim=True).values + payload.min(dim=-1, keepdim=True).values)
    return (payload > thr).to(torch.uint8)


def gather_payloads(local, equal_shards: bool = False, force: bool = False):
    """all-gather per-frame payloads [n_local, L] uint8 -> [n_total, L] in rank order.
    ``equal_shards=True`` promises every rank holds the same number of frames: one collective, no size
    exchange and no host synchronisation (the benchmark's steady state).  Otherwise ragged shards are
    handled with a size exchange and padding."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return local            # ``force``: issue the collective even in a one-rank group (dry run)
    world = dist.get_world_size()
    if equal_shards:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    n_local = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        out = torch.empty((world * sizes[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def _row_keys(rows):
    """One integer per row of a 0/1 matrix, equal exactly when the rows are equal (bits packed into a uint64);
    None when the rows do not fit (more than 64 columns, or entries other than 0/1)."""
    n, L = rows.shape
    if L > 64 or rows.dtype.kind not in "buif" or (rows.size and (rows.min() < 0 or rows.max() > 1)):
        return None
    packed = np.packbits(rows.astype(np.uint8, copy=False), axis=1)
    buf = np.zeros((n, 8), dtype=np.uint8)
    buf[:, : packed.shape[1]] = packed
    return buf.view(np.uint64).reshape(n)


def _vote_rows(rows):
    """vote() by whole-row comparison: any width, any values."""
    uniq, first, counts = np.unique(rows, axis=0, return_index=True, return_counts=True)
    best = np.lexsort((first, -counts))[0]           # highest count, then earliest first appearance
    return rows[first[best]].astype(np.int64), counts[best] / rows.shape[0]


def vote(patterns):
    """Most common whole pattern and its frequency; ties go to the pattern seen first
    (collections.Counter.most_common semantics, as the reference).  patterns: [n, L] of 0/1."""
    rows = np.asarray(patterns)
    if rows.ndim != 2 or rows.shape[0] == 0:
        return None, None
    keys = _row_keys(rows)
    if keys is None:
        return _vote_rows(np.ascontiguousarray(rows))
    _, first, counts = np.unique(keys, return_index=True, return_counts=True)
    best = np.lexsort((first, -counts))[0]
    return rows[first[best]].astype(np.int64), counts[best] / rows.shape[0]


def vote_segments(patterns, segment_ids):
    """Per-segment vote.  Returns {segment_id: (pattern, frequency)}.

    All segments are resolved in one pass (two sorts over the packed row keys): at 8 ranks x 300 frames per
    step the host has about a millisecond for this before the next step's payloads land."""
    rows = np.asarray(patterns)
    seg = np.asarray(segment_ids)
    n = rows.shape[0] if rows.ndim == 2 else 0
    keys = _row_keys(rows) if n else None
    if keys is None:
        return {int(s): vote(rows[seg == s]) for s in np.unique(seg)}
    order = np.lexsort((keys, seg))                  # by segment, then pattern; stable, so first sightings lead
    ks, ss = keys[order], seg[order]
    new_run = np.ones(n, dtype=bool)
    new_run[1:] = (ks[1:] != ks[:-1]) | (ss[1:] != ss[:-1])
    starts = np.flatnonzero(new_run)
    counts = np.diff(np.append(starts, n))
    first = order[starts]
    run_seg = ss[starts]
    pick = np.lexsort((first, -counts, run_seg))     # per segment: highest count, then earliest first appearance
    rs = run_seg[pick]
    head = np.ones(len(pick), dtype=bool)
    head[1:] = rs[1:] != rs[:-1]
    winners = pick[head]
    sizes = np.add.reduceat(counts[pick], np.flatnonzero(head))
    return {int(run_seg[w]): (rows[first[w]].astype(np.int64), counts[w] / size)
            for w, size in zip(winners, sizes)}


def group_segment_ids(segment_ids, ranks: int, steps: int):
    """Segment ids for the rows of a GROUP of ``steps`` steps gathered in one collective.

    Small shards are issued several steps per host iteration (a 48-frame segment is 0.2 ms of GPU work): every rank
    contributes its ``steps * n`` payload rows in (step, frame) order, so the all-gather returns them rank-major,
    [ranks, steps, n].  ``segment_ids`` are the ids of ONE step's rows in rank-major order ([ranks * n]); the result gives
    step g's segment s the id ``g * S + s`` (S = max id + 1), so that ONE vote_segments call resolves every step of the
    group and votes never mix steps."""
    seg = np.asarray(segment_ids)
    if steps == 1:
        return seg
    n = seg.size // max(ranks, 1)
    S = int(seg.max()) + 1 if seg.size else 1
    return (np.arange(steps)[None, :, None] * S + seg.reshape(ranks, 1, n)).reshape(-1)


def vote_groups(patterns, segment_ids, ranks: int, steps: int):
    """Per-step, per-segment votes of a gathered group: a list (one entry per step) of {segment: (pattern, frequency)}.
    ``patterns``: [ranks * steps * n, L] rank-major rows; ``segment_ids``: one step's ids ([ranks * n])."""
    seg = np.asarray(segment_ids)
    S = int(seg.max()) + 1 if seg.size else 1
    votes = vote_segments(patterns, group_segment_ids(seg, ranks, steps))
    if steps == 1:
        return [votes]
    out = [dict() for _ in range(steps)]
    for key, v in votes.items():
        out[key // S][key % S] = v
    return out


def soft_vote(soft_sums, perm, segment_ids=None):
    """Build extension (not reference semantics): combine per-frame soft sums [n, L] (offmark's
    DctEngine.detect_soft) by ADDING them over the frames of each segment, undo the key permutation and read
    each payload position by the sign of its total.  Returns {segment: payload uint8 [L]} (segment 0 if none)."""
    s = np.asarray(soft_sums, dtype=np.int64)
    seg = np.zeros(len(s), dtype=np.int64) if segment_ids is None else np.asarray(segment_ids)
    out = {}
    for k in np.unique(seg):
        total = s[seg == k].sum(axis=0)
        payload = np.empty_like(total)
        payload[np.asarray(perm)] = total
        out[int(k)] = (payload > 0).astype(np.uint8)
    return out
