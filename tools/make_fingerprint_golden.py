#!/usr/bin/env python3
"""Golden vectors for the A/B fingerprint layer, produced by the reference's own functions.

The reference's workflow scripts cannot be imported (module-level imports of cv2 / ffmpeg wrappers), but the
bookkeeping functions in them are plain Python.  This script reads those scripts from /root/reference, takes the
named function definitions out of the parsed module (ast), executes exactly those definitions in a namespace
that holds only what they use (numpy, os, json, random, a logger, pathlib) and records inputs and outputs in
tests/golden/fingerprint_layer.json.  Nothing of the reference's text is written anywhere: the fixture is data.

  python tools/make_fingerprint_golden.py            (needs /root/reference; runs on CPU)
"""
import ast
import json
import logging
import os
import random
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OFFMARK_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))


def functions_of(relpath, names, extra=None):
    """Compile the named top-level function definitions of a reference script into a fresh namespace."""
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in picked} == set(names), (relpath, names)
    ns = {"np": np, "os": os, "json": json, "random": random, "logger": logging.getLogger("reference"), "Path": Path}
    ns.update(extra or {})
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


def classes_of(relpath, names, extra=None):
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert {n.name for n in picked} == set(names), (relpath, names)
    ns = {"np": np, "logger": logging.getLogger("reference")}
    ns.update(extra or {})
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


class _Frames:            # frame_reader stand-in: hands out frame numbers as 1x1x3 "frames", then None
    def __init__(self, n):
        self.n, self.i = n, 0

    def read(self):
        if self.i >= self.n:
            return None
        self.i += 1
        return np.full((1, 1, 3), self.i - 1, dtype=np.uint8)

    def close(self):
        pass


class _Cv2:               # the vote does not depend on the colour transform: pass frames through
    COLOR_BGR2YUV = 0

    @staticmethod
    def cvtColor(a, code):
        return a


def reference_votes():
    """Cross-frame vote (a10): the reference's PatternCollectorExtractor.start() over prescribed per-frame patterns."""
    from collections import Counter
    rng = np.random.default_rng(5)
    base = rng.integers(0, 2, (4, 8))
    cases = {
        "unanimous": [base[0]] * 7,
        "majority": [base[0], base[1], base[0], base[2], base[0]],
        "tie_first_seen_wins": [base[1], base[0], base[0], base[1]],
        "tie_three_way": [base[2], base[1], base[0]],
        "late_majority": [base[3], base[2], base[2], base[3], base[2]],
        "single": [base[1]],
        "empty": [],
        "random_40": [base[i] for i in rng.integers(0, 4, 40)],
        "random_41": [base[i] for i in rng.integers(0, 3, 41)],
    }
    rows = []
    for script in ("tests/segment_mark_detect_hls.py", "tests/detect_watermarks.py"):
        ns = classes_of(script, ["PatternCollectorExtractor"], {"cv2": _Cv2, "Counter": Counter})
        for name, patterns in cases.items():
            class _Decoder:
                def decode(self, yuv):
                    return int(yuv[0, 0, 0])

            class _Degenerator:
                def degenerate(self, k):
                    return np.asarray(patterns[k])
            pattern, freq = ns["PatternCollectorExtractor"](_Frames(len(patterns)), _Decoder(), _Degenerator()).start()
            rows.append([script, name, [np.asarray(p).tolist() for p in patterns],
                         None if pattern is None else np.asarray(pattern).tolist(), freq])
    return rows


def main():
    out = {"_made_by": "tools/make_fingerprint_golden.py from the reference's own function definitions"}

    mark = functions_of("tests/mark_video_to_hls.py", ["generate_payload_for_segment"])
    hls = functions_of("tests/segment_mark_detect_hls.py", ["generate_payload_for_segment"])
    det = functions_of("tests/detect_watermarks.py", ["decode_watermark_pattern", "generate_payload_for_segment",
                                                      "extract_segment_number_from_filename", "load_payload_mappings",
                                                      "load_segment_copies"])
    leak = functions_of("tests/generate_leak.py", ["select_copies", "load_segment_copies"])

    out["payload_segment_copy"] = [[s, c, mark["generate_payload_for_segment"](s, c).tolist()]
                                   for s in list(range(0, 20)) + [31, 32, 255, 256, 1000] for c in (0, 1, 2, 3, 15, 16, 17)]
    out["payload_segment_copy_default"] = [[s, mark["generate_payload_for_segment"](s).tolist()] for s in range(0, 20)]
    out["payload_segment_only"] = [[s, hls["generate_payload_for_segment"](s).tolist()]
                                   for s in list(range(0, 20)) + [127, 128, 255, 256, 257, 1000]]
    assert all(det["generate_payload_for_segment"](s, c).tolist() == p for s, c, p in out["payload_segment_copy"])

    dec = det["decode_watermark_pattern"]
    cases = [[int(b) for b in format(v, "08b")] for v in range(256)]
    out["decode_pattern"] = [[p, list(dec(np.array(p)))] for p in cases]
    out["decode_pattern_misc"] = [
        ["none", None, list(dec(None))],
        ["list", [1, 0, 1, 0, 0, 1, 1, 0], list(dec([1, 0, 1, 0, 0, 1, 1, 0]))],
        ["short", [1, 0, 1], list(dec(np.array([1, 0, 1])))],
        ["long", [1, 0, 1, 0, 0, 1, 1, 0, 1, 1], list(dec(np.array([1, 0, 1, 0, 0, 1, 1, 0, 1, 1])))],
    ]

    out["segment_number_from_filename"] = [[n, det["extract_segment_number_from_filename"](n)] for n in
                                           ("segment_001.mp4", "/a/b/segment_017.mp4", "marked_seg3_copy1.mp4",
                                            "marked_seg012_copy2.m4s", "clip.mp4", "x_y_42.mp4", "seg7.mp4")]

    # view number -> one copy per segment (api/main.py:216-252): run the whole function against a directory that holds
    # every candidate segment file and read the chosen copies back from the playlist it returns
    with tempfile.TemporaryDirectory() as tmp:
        hls_dir = Path(tmp) / "hls"
        hls_dir.mkdir()
        for i in range(16):
            for c in range(5):
                (hls_dir / f"marked_seg{i:03d}_copy{c}.m4s").touch()
        api = functions_of("api/main.py", ["create_view_playlist"], {"PROCESSED_DIR": Path(tmp)})
        rows = []
        for copies, segments in ((2, 4), (3, 8), (3, 4), (4, 6)):
            for view in list(range(0, 30)) + [80, 81, 255, 6560, 6561]:
                text = api["create_view_playlist"](view, copies, segments)
                chosen = [[int(l.split("marked_seg")[1][:3]), int(l.split("_copy")[1].split(".")[0])]
                          for l in text.splitlines() if l.startswith("/hls/marked_seg")]
                rows.append([view, copies, segments, chosen])
        out["view_playlist_segments"] = rows

    # sidecars: written by THIS repository's writer, read back and consumed by the reference's loaders / selector
    from offmark import fingerprint as fp
    segs, ncopies = [0, 1, 2, 3, 4, 5, 6, 7], 3
    sidecars = {
        "segment_payloads": {f"{s}_{c}": fp.payload_for_segment(s, c).tolist() for s in segs for c in range(ncopies)},
        "segment_copies": {"total_segments": len(segs), "copies_per_segment": ncopies,
                           "total_marked_segments": len(segs) * ncopies,
                           "segments": {str(s): [{"file": f"marked_seg{s}_copy{c}.mp4",
                                                  "payload": fp.payload_for_segment(s, c).tolist(), "copy_index": c}
                                                 for c in range(ncopies)] for s in segs}},
        "failed_segments": [],
    }
    with tempfile.TemporaryDirectory() as tmp:
        paths = fp.write_sidecars(tmp, sidecars)
        payloads = det["load_payload_mappings"](os.path.join(tmp, "segment_payloads.json"))
        copies_info = leak["load_segment_copies"](os.path.join(tmp, "segment_copies.json"))
        assert det["load_segment_copies"](os.path.join(tmp, "segment_copies.json")) == copies_info
        assert payloads == sidecars["segment_payloads"] and copies_info == sidecars["segment_copies"]
        sel = []
        for pattern in ("01201201", "00000000", "22222222", "98765432", "0120120199"):
            files, chosen = leak["select_copies"](copies_info, os.path.join(tmp, "segment_copies.json"), pattern=pattern)
            sel.append([pattern, chosen, [os.path.basename(f) for f in files]])
        try:
            leak["select_copies"](copies_info, os.path.join(tmp, "segment_copies.json"), pattern="012")
            short = "no error"
        except ValueError as exc:
            short = str(exc)
        out["select_copies"] = {"segments": segs, "copies": ncopies, "cases": sel, "too_short_message": short,
                                "files_written": sorted(os.path.basename(p) for p in paths)}

    out["cross_frame_vote"] = reference_votes()

    dst = os.path.join(ROOT, "tests", "golden", "fingerprint_layer.json")
    with open(dst, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
