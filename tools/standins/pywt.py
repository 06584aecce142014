"""Minimal ``pywt`` stand-in used ONLY by tools/make_golden.py: one-level 2-D Haar from the oracle's
restated primitives (PyWavelets is not installed; its float arithmetic is parity-unpinned)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import offmark_oracle as _o  # noqa: E402


def dwt2(data, wavelet):
    assert wavelet == "haar"
    return _o.haar_dwt2(data)


def idwt2(coeffs, wavelet):
    assert wavelet == "haar"
    return _o.haar_idwt2(coeffs)
