"""Second, INDEPENDENT ``pywt`` stand-in for tools/make_golden.py --standin scipy: the one-level 2-D Haar transform from
its closed form on 2x2 pixel groups (cA = (a + b + c + d) / 2, ...), float32.  Nothing here comes from the oracle, whose
form applies 1/sqrt(2) along one axis and then the other.  The reference changes cA only (dwt_dct_svd_encoder.py:27-44),
so the sign convention of the detail bands only has to agree between dwt2 and idwt2."""
import numpy as np

_F = np.float32
_H = _F(0.5)


def dwt2(data, wavelet):
    assert wavelet == "haar"
    x = np.asarray(data, dtype=_F)
    assert x.shape[-2] % 2 == 0 and x.shape[-1] % 2 == 0
    a, b = x[..., 0::2, 0::2], x[..., 0::2, 1::2]
    c, d = x[..., 1::2, 0::2], x[..., 1::2, 1::2]
    ca = ((a + b) + (c + d)) * _H
    ch = ((a + b) - (c + d)) * _H
    cv = ((a - b) + (c - d)) * _H
    cd = ((a - b) - (c - d)) * _H
    return ca, (ch, cv, cd)


def idwt2(coeffs, wavelet):
    assert wavelet == "haar"
    ca, (ch, cv, cd) = coeffs
    ca, ch, cv, cd = (np.asarray(t, dtype=_F) for t in (ca, ch, cv, cd))
    out = np.empty(ca.shape[:-2] + (ca.shape[-2] * 2, ca.shape[-1] * 2), _F)
    out[..., 0::2, 0::2] = ((ca + ch) + (cv + cd)) * _H
    out[..., 0::2, 1::2] = ((ca + ch) - (cv + cd)) * _H
    out[..., 1::2, 0::2] = ((ca - ch) + (cv - cd)) * _H
    out[..., 1::2, 1::2] = ((ca - ch) - (cv - cd)) * _H
    return out
