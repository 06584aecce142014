"""ctypes binding of liboffmark_hip.so (C ABI declared in include/offmark_hip.h).

The shared library holds the hand-written gfx950 kernels; it is built in-tree by
``__graft_entry__.build()`` (``hipcc --offload-arch=gfx950``).  There is no CPU fallback: if the
library is missing, or no GPU is visible when a compute entry point is called, this module
raises.  torch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "liboffmark_hip.so")

#: every symbol include/offmark_hip.h declares
SYMBOLS = (
    "ofmk_version", "ofmk_last_error", "ofmk_workspace_bytes", "ofmk_embed_rgb8", "ofmk_detect_rgb8",
    "ofmk_embed_detect_rgb8", "ofmk_encode_yuv32f", "ofmk_decode_yuv32f", "ofmk_debug_planes",
    "ofmk_stage_analyze_rgb8", "ofmk_stage_mark_rgb8", "ofmk_hbm_copy", "ofmk_set_fused_verify",
    "ofmk_timing_enable", "ofmk_timing_collect", "ofmk_timing_disable", "ofmk_payloads_from_counts",
    "ofmk_svd_embed_rgb8", "ofmk_svd_detect_rgb8", "ofmk_svd_embed_detect_rgb8", "ofmk_svd_encode_yuv32f",
    "ofmk_svd_decode_yuv32f", "ofmk_set_onepass_grid", "ofmk_onepass_error", "ofmk_detect_soft_rgb8",
)


class HipLibraryMissing(RuntimeError):
    pass


class HipError(RuntimeError):
    pass


_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the library once and declare signatures.  Raises HipLibraryMissing if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryMissing(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  offmark's DCT codec has no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    vp, i32, f64, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    lib.ofmk_version.restype = i32
    lib.ofmk_last_error.restype = C.c_char_p
    lib.ofmk_workspace_bytes.restype = sz
    lib.ofmk_workspace_bytes.argtypes = [i32, i32, i32]
    lib.ofmk_embed_rgb8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, f64, i32, vp, sz, vp]
    lib.ofmk_detect_rgb8.argtypes = [vp, i32, i32, i32, i32, f64, vp, vp, i32, vp, sz, vp]
    lib.ofmk_embed_detect_rgb8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, f64, i32, vp, vp, i32, vp, sz, vp]
    lib.ofmk_encode_yuv32f.argtypes = [vp, i32, i32, i32, vp, i32, vp, f64, i32, vp, sz, vp]
    lib.ofmk_decode_yuv32f.argtypes = [vp, i32, i32, i32, i32, f64, vp, vp, i32, vp, sz, vp]
    lib.ofmk_debug_planes.argtypes = [vp, i32, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.ofmk_stage_analyze_rgb8.argtypes = [vp, i32, i32, i32, vp, sz, vp]
    lib.ofmk_stage_mark_rgb8.argtypes = [vp, vp, i32, i32, i32, vp, f64, i32, vp, sz, vp]
    lib.ofmk_hbm_copy.argtypes = [vp, vp, sz, vp]
    lib.ofmk_set_fused_verify.argtypes = [i32]
    lib.ofmk_set_fused_verify.restype = None
    lib.ofmk_payloads_from_counts.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.ofmk_payloads_from_counts.restype = i32
    lib.ofmk_svd_embed_rgb8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, f64, vp]
    lib.ofmk_svd_detect_rgb8.argtypes = [vp, i32, i32, i32, i32, f64, vp, vp, vp]
    lib.ofmk_svd_embed_detect_rgb8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, f64, i32, vp, vp, vp]
    lib.ofmk_svd_encode_yuv32f.argtypes = [vp, i32, i32, i32, vp, i32, vp, f64, vp]
    lib.ofmk_svd_decode_yuv32f.argtypes = [vp, i32, i32, i32, f64, vp, vp]
    for name in ("ofmk_svd_embed_rgb8", "ofmk_svd_detect_rgb8", "ofmk_svd_embed_detect_rgb8",
                 "ofmk_svd_encode_yuv32f", "ofmk_svd_decode_yuv32f"):
        getattr(lib, name).restype = i32
    lib.ofmk_detect_soft_rgb8.argtypes = [vp, i32, i32, i32, i32, f64, vp, i32, vp, sz, vp]
    lib.ofmk_detect_soft_rgb8.restype = i32
    lib.ofmk_set_onepass_grid.argtypes = [i32]
    lib.ofmk_set_onepass_grid.restype = None
    lib.ofmk_onepass_error.argtypes = [vp, sz, i32, i32, i32, C.POINTER(C.c_uint)]
    lib.ofmk_onepass_error.restype = i32
    lib.ofmk_timing_enable.argtypes = [i32, C.c_uint]
    lib.ofmk_timing_enable.restype = i32
    lib.ofmk_timing_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.ofmk_timing_collect.restype = i32
    lib.ofmk_timing_disable.argtypes = []
    lib.ofmk_timing_disable.restype = None
    for name in ("ofmk_embed_rgb8", "ofmk_detect_rgb8", "ofmk_embed_detect_rgb8", "ofmk_encode_yuv32f",
                 "ofmk_decode_yuv32f", "ofmk_debug_planes", "ofmk_stage_analyze_rgb8", "ofmk_stage_mark_rgb8",
                 "ofmk_hbm_copy"):
        getattr(lib, name).restype = i32
    if lib.ofmk_version() != 1:
        raise HipError(f"ABI version mismatch: library reports {lib.ofmk_version()}, binding expects 1")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().ofmk_last_error().decode("utf-8", "replace")
        raise HipError(f"offmark HIP call failed (code {rc}): {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise HipError("no HIP device visible: offmark's DCT codec runs only on the GPU (no CPU fallback)")
    return torch


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
