"""Minimal ``cv2`` stand-in used ONLY by tools/make_golden.py (fixture generation, build
container).  OpenCV is not installed here; the three functions the reference's DCT path
calls are supplied from the oracle's restated primitives, so golden vectors pin the
reference's control flow and scalar semantics, not OpenCV's float arithmetic
("parity unpinned" for that part -- see oracle/offmark_oracle.py header)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import offmark_oracle as _o  # noqa: E402

COLOR_BGR2YUV = 82
COLOR_YUV2BGR = 84


def dct(src):
    return _o.dct4x4(src) if src.shape == (4, 4) else _o.dct8x8(src)


def idct(src):
    return _o.idct4x4(src) if src.shape == (4, 4) else _o.idct8x8(src)


def cvtColor(src, code):
    if code == COLOR_BGR2YUV:
        return _o.bgr2yuv_f32(src)
    if code == COLOR_YUV2BGR:
        return _o.yuv2bgr_f32(src)
    raise NotImplementedError(code)
