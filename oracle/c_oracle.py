"""ctypes loader of oracle/_build/liboffmark_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboffmark_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "offmark_oracle.c")):
            build()
        lib = C.CDLL(LIB)
        u8p, i64p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        lib.ofo_mark_frames.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, i64p, C.c_double, C.c_int, C.c_int]
        lib.ofo_check_frames.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, f64p, C.c_int]
        lib.ofo_mark_frames.restype = lib.ofo_check_frames.restype = C.c_int
        _lib = lib
    return _lib


def mark_frames(frames, wm, alpha=20, legacy=True, threads=1):
    """frames u8 [n,H,W,3]; wm 0/1 ints with at least (H//8)*(W//8) entries.  Returns (marked, threads used)."""
    lib = load()
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, H, W, _ = frames.shape
    wm = np.ascontiguousarray(np.asarray(wm).reshape(-1), dtype=np.int64)
    out = np.empty_like(frames)
    used = lib.ofo_mark_frames(frames.ctypes.data_as(C.POINTER(C.c_uint8)), out.ctypes.data_as(C.POINTER(C.c_uint8)), n, H, W,
                               wm.ctypes.data_as(C.POINTER(C.c_int64)), float(alpha), int(legacy), int(threads))
    return out, used


def check_frames(frames, alpha=20, legacy=True, threads=1):
    """Returns (bits float64 [n, H*W//64], threads used)."""
    lib = load()
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, H, W, _ = frames.shape
    bits = np.empty((n, H * W // 64), dtype=np.float64)
    used = lib.ofo_check_frames(frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, H, W, float(alpha), int(legacy),
                                bits.ctypes.data_as(C.POINTER(C.c_double)), int(threads))
    return bits, used
