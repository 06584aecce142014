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

ABI_VERSION = 6


class Opts(C.Structure):
    """ofmk_opts: per-call options (flags, optional timing object).  None / NULL = defaults.
    An Opts made by Timing.opts() keeps its Timing alive (``_timing``) and is invalidated by Timing.close(): its
    pointer is nulled, so an engine that still holds it simply launches without events instead of touching freed memory."""
    _fields_ = [("flags", C.c_uint32), ("xcds", C.c_uint32), ("timing", C.c_void_p)]
    _timing = None


F_SEPARATE_DETECT = 1
F_LINEAR_TILES = 2          # tile order of the frame-writing DCT kernel: force workgroup index order ...
F_XCD_TILES = 4             # ... or the XCD-aware order; neither: the library's static rule on the launch size (offmark_hip.h)
F_PARTIAL_COUNTS = 8        # DwtDctSvd read-outs: counts = per-workgroup partial sums [n][tiles][L], stored not added (no fill dispatch)
XCD_TILES_MIN_BYTES = 192 * 1080 * 1920 * 3
YUV_I420, YUV_NV12 = 0, 1
TIMING_KINDS = ("analyze", "finalize", "mark", "mark_fused", "svd", "planar_analyze", "planar_mark")

_vp, _i32, _f64, _sz, _u32 = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_uint
_op = C.POINTER(Opts)
_dp = C.POINTER(C.c_double)

#: every symbol include/offmark_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ofmk_version": (_i32, []),
    "ofmk_last_error": (C.c_char_p, []),
    "ofmk_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "ofmk_embed_rgb8": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _f64, _i32, _vp, _sz, _vp, _op]),
    "ofmk_detect_rgb8": (_i32, [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _vp, _i32, _vp, _sz, _vp, _op]),
    "ofmk_detect_soft_rgb8": (_i32, [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _i32, _vp, _sz, _vp, _op]),
    "ofmk_embed_detect_rgb8": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _f64, _i32, _vp, _vp, _i32, _vp, _sz,
                                      _vp, _op]),
    "ofmk_encode_yuv32f": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _f64, _i32, _vp, _sz, _vp, _op]),
    "ofmk_decode_yuv32f": (_i32, [_vp, _i32, _i32, _i32, _i32, _f64, _vp, _vp, _i32, _vp, _sz, _vp, _op]),
    "ofmk_debug_planes": (_i32, [_vp, _i32, _i32, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _op]),
    "ofmk_stage_analyze_rgb8": (_i32, [_vp, _i32, _i32, _i32, _vp, _sz, _vp, _op]),
    "ofmk_stage_mark_rgb8": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _f64, _i32, _vp, _sz, _vp, _op]),
    "ofmk_svd_embed_rgb8": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _dp, _i32, _vp, _op]),
    "ofmk_svd_detect_rgb8": (_i32, [_vp, _i32, _i32, _i32, _i32, _dp, _i32, _vp, _vp, _vp, _op]),
    "ofmk_svd_embed_detect_rgb8": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _dp, _i32, _i32, _vp, _vp, _vp, _op]),
    "ofmk_svd_encode_yuv32f": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _dp, _i32, _vp, _op]),
    "ofmk_svd_decode_yuv32f": (_i32, [_vp, _i32, _i32, _i32, _dp, _i32, _vp, _vp, _op]),
    "ofmk_payloads_from_counts": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _op]),
    "ofmk_svd_count_tiles": (_i32, [_i32, _i32, _i32]),
    "ofmk_payloads_from_partial_counts": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _op]),
    "ofmk_embed_yuv420": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _f64, _i32, _vp, _sz, _vp, _op]),
    "ofmk_detect_yuv420": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _f64, _vp, _vp, _i32, _vp, _sz, _vp, _op]),
    "ofmk_embed_detect_yuv420": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _f64, _i32, _vp, _vp, _i32, _vp,
                                        _sz, _vp, _op]),
    "ofmk_yuv420_to_rgb8": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _op]),
    "ofmk_rgb8_to_yuv420": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _op]),
    "ofmk_probe_xcc": (_i32, [_vp, _i32, _vp, _op]),
    "ofmk_hbm_copy": (_i32, [_vp, _vp, _sz, _vp]),
    "ofmk_hbm_read": (_i32, [_vp, _sz, _vp, _vp]),
    "ofmk_timing_create": (_i32, [_i32, _u32, C.POINTER(_vp)]),
    "ofmk_timing_collect": (_i32, [_vp, C.POINTER(_f64), C.POINTER(_i32)]),
    "ofmk_timing_durations": (_i32, [_vp, C.POINTER(C.c_float), C.POINTER(_i32), _i32]),
    "ofmk_timing_destroy": (None, [_vp]),
}
SYMBOLS = tuple(SIGNATURES)


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
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.ofmk_version() != ABI_VERSION:
        raise HipError(f"ABI version mismatch: library reports {lib.ofmk_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


class Timing:
    """Caller-owned event pool (ofmk_timing_create): pass `.opts()` to the calls whose kernels should be timed."""

    def __init__(self, max_launches: int, kind_mask: int = 0):
        self.lib = load()
        self.handle = _vp()
        self._handed_out = []
        check(self.lib.ofmk_timing_create(int(max_launches), int(kind_mask), C.byref(self.handle)))

    def opts(self, flags: int = 0) -> Opts:
        import weakref
        o = Opts(flags, 0, self.handle)
        o._timing = self                                  # the pool lives at least as long as the options that name it
        self._handed_out.append(weakref.ref(o))
        return o

    def collect(self) -> dict:
        ms = (_f64 * len(TIMING_KINDS))()
        cnt = (_i32 * len(TIMING_KINDS))()
        check(self.lib.ofmk_timing_collect(self.handle, ms, cnt))
        return {k: dict(ms_total=ms[i], launches=cnt[i]) for i, k in enumerate(TIMING_KINDS)}

    def durations(self, cap: int = 4096):
        """[(ms, kind name)] of the recorded launches in launch order; does not rewind the pool (call before collect())."""
        ms = (C.c_float * cap)()
        kinds = (_i32 * cap)()
        n = self.lib.ofmk_timing_durations(self.handle, ms, kinds, cap)
        if n < 0:
            check(n)
        return [(float(ms[i]), TIMING_KINDS[kinds[i]] if 0 <= kinds[i] < len(TIMING_KINDS) else "?") for i in range(n)]

    def close(self):
        for ref in self._handed_out:                      # options still held elsewhere stop naming the pool
            o = ref()
            if o is not None:
                o.timing = None
        self._handed_out = []
        if self.handle:
            self.lib.ofmk_timing_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scales3(scale=15, scales=None):
    """ctypes double[3] for the DwtDctSvd entry points: `scales` = per-channel list as in
    DwtDctSvdEncoder(scales=[0,15,0]), or the single channel-1 `scale`."""
    v = [0.0, float(scale), 0.0] if scales is None else [float(x) for x in scales]
    if len(v) != 3:
        raise ValueError("scales needs three entries (one per YUV channel)")
    return (C.c_double * 3)(*v)


def opts_ref(o):
    """ctypes argument for an optional Opts."""
    return None if o is None else C.byref(o)


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
