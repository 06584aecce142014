"""The A/B fingerprint layer (offmark.fingerprint) against vectors produced by the reference's own
functions (tools/make_fingerprint_golden.py ran them in the build container; the fixture holds inputs and outputs)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def g():
    return json.load(open(os.path.join(GOLDEN, "fingerprint_layer.json")))


def test_payload_schemes(g):
    from offmark import fingerprint as fp
    for s, c, bits in g["payload_segment_copy"]:            # tests/mark_video_to_hls.py:27-43
        assert fp.payload_for_segment(s, c).tolist() == bits, (s, c)
    for s, bits in g["payload_segment_copy_default"]:       # same function, copy_index left at its default 0
        assert fp.payload_for_segment(s, 0).tolist() == bits
    for s, bits in g["payload_segment_only"]:               # tests/segment_mark_detect_hls.py:42-55
        assert fp.payload_for_segment(s).tolist() == bits, s


def test_decode_pattern(g):
    from offmark import fingerprint as fp
    for bits, expected in g["decode_pattern"]:              # tests/detect_watermarks.py:145-172, all 256 patterns
        assert list(fp.decode_pattern(np.array(bits))) == expected
    for _name, bits, expected in g["decode_pattern_misc"]:  # None, a list, fewer than / more than 8 bits
        assert list(fp.decode_pattern(bits)) == expected, _name


def test_view_number_to_copies(g):
    from offmark import fingerprint as fp
    for view, copies, segments, chosen in g["view_playlist_segments"]:   # api/main.py:216-252
        digits = fp.view_to_copies(view, copies, segments)
        assert [[i, c] for i, c in enumerate(digits)] == chosen, (view, copies, segments)


def test_select_copies_and_sidecars(g, tmp_path):
    from offmark import fingerprint as fp
    sc = g["select_copies"]
    for pattern, chosen, files in sc["cases"]:              # tests/generate_leak.py:59-108
        mine = fp.select_copies(pattern, len(sc["segments"]), sc["copies"])
        assert mine == chosen
        assert [f"marked_seg{s}_copy{c}.mp4" for s, c in zip(sc["segments"], mine)] == files
    with pytest.raises(ValueError) as err:
        fp.select_copies("012", len(sc["segments"]), sc["copies"])
    assert str(err.value) == sc["too_short_message"]
    # the writer leaves out failed_segments.json when nothing failed, as the reference does (mark_video_to_hls.py:418-427)
    paths = fp.write_sidecars(str(tmp_path), {"segment_payloads": {}, "segment_copies": {"segments": {}}, "failed_segments": []})
    assert sorted(os.path.basename(p) for p in paths) == sc["files_written"]


def test_segment_number_from_filename(g):
    from offmark import fingerprint as fp
    for name, expected in g["segment_number_from_filename"]:   # tests/detect_watermarks.py:50-80
        assert fp.segment_number_from_filename(name) == expected, name


def test_cross_frame_vote_against_the_reference_collector(g):
    """a10: PatternCollectorExtractor.start() of the reference (segment_mark_detect_hls.py:119-155,
    detect_watermarks.py:101-137) run on prescribed per-frame patterns, ties and the empty case included."""
    from offmark.dist.vote import vote, vote_segments
    for _script, name, patterns, winner, freq in g["cross_frame_vote"]:
        rows = np.asarray(patterns, dtype=np.int64).reshape(len(patterns), -1) if patterns else np.zeros((0, 8), np.int64)
        pattern, f = vote(rows)
        if winner is None:
            assert pattern is None and f is None, name
            continue
        assert pattern.tolist() == winner and f == freq, name
        # the all-segments-at-once form agrees, whatever else shares the batch
        both = vote_segments(np.concatenate([rows, rows[::-1]]), np.repeat([3, 9], len(rows)))
        assert both[3][0].tolist() == winner and both[3][1] == freq, name
