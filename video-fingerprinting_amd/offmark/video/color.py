"""Host-side float32 colour transform used by the GENERIC (per-frame, any-encoder) pipeline path.

Same formulas as OpenCV's float COLOR_BGR2YUV / COLOR_YUV2BGR (delta = 0.5; channel 0 is "B"
whatever the caller stored there), which the reference calls at video/embedder.py:34,36 and
video/extractor.py:31.  The HIP codecs never use this: they transform on the device."""
import numpy as np

_F = np.float32


def bgr2yuv(img):
    a = np.asarray(img, dtype=_F)
    c0, c1, c2 = a[..., 0], a[..., 1], a[..., 2]
    y = c0 * _F(0.114) + (c1 * _F(0.587) + c2 * _F(0.299))
    u = (c0 - y) * _F(0.492) + _F(0.5)
    v = (c2 - y) * _F(0.877) + _F(0.5)
    return np.stack([y, u, v], axis=-1)


def yuv2bgr(img):
    a = np.asarray(img, dtype=_F)
    y, u, v = a[..., 0], a[..., 1] - _F(0.5), a[..., 2] - _F(0.5)
    return np.stack([y + u * _F(2.032), y + u * _F(-0.395) + v * _F(-0.581), y + v * _F(1.140)], axis=-1)
