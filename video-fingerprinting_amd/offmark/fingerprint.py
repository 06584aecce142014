"""A/B-segment fingerprint bookkeeping on top of the per-frame payloads (host-side, integers only).

Restates the payload conventions of the reference's workflow scripts so that segments marked here are
read by its detector and vice versa:
  * segment payload, 8 bits of the segment number  -- tests/segment_mark_detect_hls.py:42-55
  * segment(4 bits) || copy(4 bits) payload         -- tests/mark_video_to_hls.py:27-43
  * pattern -> (segment, copy)                      -- tests/detect_watermarks.py:145-172
  * leak pattern -> one copy per segment            -- tests/generate_leak.py:59-108 (the selection rule)
  * view number -> base-C digit string              -- api/main.py:220-230
  * per-segment detection -> copy sequence          -- tests/detect_watermarks.py:345-364 (no-mapping branch)
  * segment file name -> segment number             -- tests/detect_watermarks.py:50-80
Pinned by tests/golden/fingerprint_layer.json: inputs/outputs of the reference's own functions
(tools/make_fingerprint_golden.py).
"""
from __future__ import annotations

import numpy as np


def payload_for_segment(segment_number: int, copy_index: int | None = None) -> np.ndarray:
    """8 payload bits, MSB first.  With ``copy_index``: 4 bits of segment % 16 then 4 bits of copy % 16."""
    if copy_index is None:
        text = format(segment_number % 256, "08b")
    else:
        text = format(segment_number % 16, "04b") + format(copy_index % 16, "04b")
    return np.array([int(b) for b in text])


def decode_pattern(pattern):
    """(segment_number, copy_index) from at least 8 bits, or (None, None)."""
    if pattern is None:
        return None, None
    bits = [int(b) for b in np.asarray(pattern).reshape(-1)]
    if len(bits) < 8:
        return None, None
    return int("".join(map(str, bits[:4])), 2), int("".join(map(str, bits[4:8])), 2)


def select_copies(pattern: str, num_segments: int, num_copies: int) -> list[int]:
    """Copy index used for each segment of a leak: digit i of ``pattern`` modulo the number of copies."""
    if len(pattern) < num_segments:
        raise ValueError(f"Pattern '{pattern}' is too short for {num_segments} segments")
    return [int(pattern[i]) % num_copies for i in range(num_segments)]


def view_to_copies(view_number: int, num_copies: int, num_segments: int) -> list[int]:
    """Base-``num_copies`` digits of the view number, most significant first, zero-padded to the segment count."""
    digits = []
    v = view_number
    while v > 0:
        digits.append(v % num_copies)
        v //= num_copies
    while len(digits) < num_segments:
        digits.append(0)
    digits.reverse()
    return digits


def segment_number_from_filename(filename: str):
    """Segment number of a segment file (tests/detect_watermarks.py:50-80): the first '_'-separated part of the
    basename that is all digits, else the first run of digits anywhere in the basename, else None."""
    import os
    import re
    base = os.path.basename(filename)
    for part in base.split("_"):
        if part.isdigit():
            return int(part)
    m = re.search(r"(\d+)", base)
    return int(m.group(1)) if m else None


def identify_copies(segment_votes: dict, segment_numbers=None) -> list[int | None]:
    """Copy sequence of a leaked stream from per-segment votes {segment: (pattern, frequency)}.
    A segment whose decoded segment field does not match its own number (mod 16) yields None."""
    keys = sorted(segment_votes) if segment_numbers is None else list(segment_numbers)
    out = []
    for s in keys:
        pattern, _freq = segment_votes[s]
        seg, copy = decode_pattern(pattern)
        out.append(copy if seg is not None and seg == s % 16 else None)
    return out


# ---------------------------------------------------------------------------------------------
# Marking N copies per segment, verifying them, and the reference's JSON sidecars
# (tests/mark_video_to_hls.py:330-434).  The pixel work is one batched GPU call per copy.
# ---------------------------------------------------------------------------------------------

def mark_segment_copies(encoder, decoder, frames, segment_of_frame, num_copies: int, key=0, min_frequency: float = 0.5):
    """Mark ``num_copies`` versions of every segment and verify each one.

    encoder / decoder: HIP codecs offering ``encode_frames_u8`` / ``decode_frames_u8`` (DctEncoder+DctDecoder or
    DwtDctSvdEncoder+DwtDctSvdDecoder).  frames: CUDA uint8 [n, H, W, 3]; segment_of_frame: int array [n].
    Returns (copies, sidecars): copies[c] is the marked tensor of copy c; sidecars holds the dicts the
    reference writes as segment_payloads.json / segment_copies.json / failed_segments.json, with the same
    keys.  A copy fails verification when its per-segment vote differs from its payload or the winning
    pattern covers fewer than ``min_frequency`` of the frames (mark_video_to_hls.py:381)."""
    import torch
    from .degenerator.de_shuffler import DeShuffler
    from .dist.vote import vote_segments
    from .generator.shuffler import Shuffler

    seg = np.asarray(segment_of_frame)
    segments = [int(s) for s in np.unique(seg)]
    n, H, W, _ = frames.shape
    N = H * W // 64
    gen = Shuffler(key=key)
    deg = DeShuffler(key=key).set_shape((8,))
    index = {(s, c): i for i, (s, c) in enumerate((s, c) for s in segments for c in range(num_copies))}
    table = np.stack([gen.generate_wm(payload_for_segment(s, c), (N,)) for s in segments for c in range(num_copies)])
    table_dev = torch.from_numpy(table.astype(np.uint8)).to(frames.device)
    copies, segment_payloads, failed = [], {}, []
    segment_copies = {str(s): [] for s in segments}
    for c in range(num_copies):
        rows = np.array([index[(int(s), c)] for s in seg], dtype=np.int32)
        marked = encoder.encode_frames_u8(frames, wm_rows=torch.from_numpy(rows).to(frames.device), wm_table=table_dev)
        counts, _ = decoder.decode_frames_u8(marked, 8)
        n_bits = decoder.bits_per_frame(H, W) if hasattr(decoder, "bits_per_frame") else N      # DwtDctSvd(blk=8): H*W//256
        votes = vote_segments(deg.degenerate_counts(counts.cpu().numpy(), n_bits), seg)
        copies.append(marked)
        for s in segments:
            payload = payload_for_segment(s, c).tolist()
            name = f"marked_seg{s}_copy{c}.mp4"
            segment_payloads[f"{s}_{c}"] = payload
            segment_copies[str(s)].append({"file": name, "payload": payload, "copy_index": c})
            pattern, freq = votes[s]
            if pattern is None or pattern.tolist() != payload or freq < min_frequency:
                failed.append({"segment": name, "segment_number": s, "copy_index": c, "expected_pattern": payload,
                               "detected_pattern": None if pattern is None else pattern.tolist(), "frequency": freq})
    sidecars = {
        "segment_payloads": segment_payloads,
        "segment_copies": {"total_segments": len(segments), "copies_per_segment": num_copies,
                           "total_marked_segments": len(segments) * num_copies, "segments": segment_copies},
        "failed_segments": failed,
    }
    return copies, sidecars


def write_sidecars(directory: str, sidecars: dict) -> list[str]:
    """Write segment_payloads.json, segment_copies.json and (if any) failed_segments.json; returns the paths."""
    import json
    import os
    os.makedirs(directory, exist_ok=True)
    paths = []
    for name in ("segment_payloads", "segment_copies", "failed_segments"):
        if name == "failed_segments" and not sidecars[name]:
            continue
        path = os.path.join(directory, name + ".json")
        with open(path, "w") as f:
            json.dump(sidecars[name], f, indent=2)
        paths.append(path)
    return paths


def identify_copies_with_payloads(segment_votes: dict, segment_payloads: dict, max_copies: int) -> list[dict]:
    """The mapping branch of detect_watermarks.py:329-344, with the segment decoded once instead of once
    per candidate copy: a copy matches when the segment's winning pattern equals that copy's recorded
    payload; among matches the highest frequency wins.  Returns one dict per segment with the keys of the
    reference's detection_results.json rows."""
    rows = []
    for s in sorted(segment_votes):
        pattern, freq = segment_votes[s]
        detected, best = None, 0
        for c in range(max_copies):
            expected = segment_payloads.get(f"{s}_{c}")
            if expected is None or pattern is None:
                continue
            if list(np.asarray(pattern).tolist()) == list(expected) and freq > best:
                best, detected = freq, c
        rows.append({"segment_number": int(s), "detected_copy_index": detected, "match_frequency": best,
                     "success": detected is not None})
    return rows
