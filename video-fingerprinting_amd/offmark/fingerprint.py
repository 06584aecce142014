"""A/B-segment fingerprint bookkeeping on top of the per-frame payloads (host-side, integers only).

Restates the payload conventions of the reference's workflow scripts so that segments marked here are
read by its detector and vice versa:
  * segment payload, 8 bits of the segment number  -- tests/segment_mark_detect_hls.py:42-55
  * segment(4 bits) || copy(4 bits) payload         -- tests/mark_video_to_hls.py:27-43
  * pattern -> (segment, copy)                      -- tests/detect_watermarks.py:145-172
  * leak pattern -> one copy per segment            -- tests/generate_leak.py:59-108 (the selection rule)
  * view number -> base-C digit string              -- api/main.py:220-230
  * per-segment detection -> copy sequence          -- tests/detect_watermarks.py:345-364 (no-mapping branch)
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
