"""Second, INDEPENDENT ``cv2`` stand-in for tools/make_golden.py --standin scipy (build container only).

Nothing here comes from the oracle: the DCTs are SciPy's (``scipy.fft.dctn`` / ``idctn``, ``norm="ortho"``, float32 in
and out -- pocketfft's arithmetic, not the oracle's butterflies) and ``cvtColor`` is OpenCV's documented float formula
written in plain NumPy float32 (separate multiply and add roundings, no fused multiply-adds).  Vectors produced with
this stand-in therefore pin the reference's logic with a primitive the oracle did not supply; what stays unpinned is
OpenCV's own float rounding (it is not installed here).  Call sites: dct_encoder.py:29,37,50,79, dct_decoder.py:23,38,66,
video/embedder.py:34,36, dwt_dct_svd_encoder.py / dwt_dct_svd_decoder.py (4x4 and 8x8 blocks)."""
import numpy as np
from scipy import fft as _fft

COLOR_BGR2YUV = 82
COLOR_YUV2BGR = 84

_F = np.float32


def dct(src):
    a = np.ascontiguousarray(src, dtype=_F)
    return _fft.dctn(a, type=2, norm="ortho").astype(_F)


def idct(src):
    a = np.ascontiguousarray(src, dtype=_F)
    return _fft.idctn(a, type=2, norm="ortho").astype(_F)


def cvtColor(src, code):
    a = np.asarray(src, dtype=_F)
    out = np.empty_like(a)
    if code == COLOR_BGR2YUV:          # Y = .114 B + .587 G + .299 R; U = .492 (B - Y) + delta; V = .877 (R - Y) + delta; delta = 0.5 for float
        b, g, r = a[..., 0], a[..., 1], a[..., 2]
        y = _F(0.114) * b + _F(0.587) * g + _F(0.299) * r
        out[..., 0] = y
        out[..., 1] = (b - y) * _F(0.492) + _F(0.5)
        out[..., 2] = (r - y) * _F(0.877) + _F(0.5)
        return out
    if code == COLOR_YUV2BGR:          # B = Y + 2.032 (U - delta); G = Y - .395 (U - delta) - .581 (V - delta); R = Y + 1.140 (V - delta)
        y, u, v = a[..., 0], a[..., 1] - _F(0.5), a[..., 2] - _F(0.5)
        out[..., 0] = y + _F(2.032) * u
        out[..., 1] = y - _F(0.395) * u - _F(0.581) * v
        out[..., 2] = y + _F(1.140) * v
        return out
    raise NotImplementedError(code)
