"""An INDEPENDENT pin of the oracle's third-party primitives (VERDICT r4 missing 3 / next 6).

The golden vectors cannot catch a transposed or mis-indexed DCT: the `cv2` stand-in that produced them IS the oracle's DCT
(tools/make_golden.py), and orthonormality / energy / inverse checks pass for a transposed transform as well.  scipy.fft is a
separate implementation of the same published definitions (orthonormal DCT-II / DCT-III, what `cv2.dct` / `cv2.idct` compute on
CV_32F blocks), so agreement here pins index order and normalisation:

  cv2.dct / cv2.idct call sites   src/offmark/embed/dct_encoder.py:29,37,50,79   extract/dct_decoder.py:23,38,66
                                  embed/dwt_dct_svd_encoder.py:43,45             extract/dwt_dct_svd_decoder.py:34
  the coefficient the codec moves dct_encoder.py:33-35  (`[2][1]`: vertical frequency 2, horizontal frequency 1)
  pywt.dwt2 / idwt2 ('haar')      embed/dwt_dct_svd_encoder.py:29-31,36-40       extract/dwt_dct_svd_decoder.py:27

OpenCV's and PyWavelets' own float ROUNDING stays parity-unpinned (neither is installed); what is pinned here is the mathematics
the stand-ins claim to restate, to 1e-4 on 0..255-scale data (float32 rounding of the results is ~3e-5 there).
"""
import numpy as np
import pytest

import offmark_oracle as orc

sfft = pytest.importorskip("scipy.fft")

TOL = 1e-4


def _blocks(n, k, seed):
    return np.random.default_rng(seed).uniform(0, 255, size=(n, k, k)).astype(np.float32)


def test_dct8x8_and_idct8x8_match_scipy_dctn():
    b = _blocks(200, 8, 11)
    want = sfft.dctn(b.astype(np.float64), type=2, norm="ortho", axes=(-2, -1))
    got = orc.dct8x8(b)
    assert got.dtype == np.float32 and np.max(np.abs(got - want)) < TOL * 8       # DC reaches 2040: scale the bound with it
    assert np.max(np.abs(got[:, 1:, 1:] - want[:, 1:, 1:])) < TOL
    c = np.random.default_rng(12).uniform(-300, 300, size=(200, 8, 8)).astype(np.float32)
    back = sfft.idctn(c.astype(np.float64), type=2, norm="ortho", axes=(-2, -1))
    assert np.max(np.abs(orc.idct8x8(c) - back)) < TOL


def test_dct4x4_and_idct4x4_match_scipy_dctn():
    b = _blocks(200, 4, 13)
    want = sfft.dctn(b.astype(np.float64), type=2, norm="ortho", axes=(-2, -1))
    assert np.max(np.abs(orc.dct4x4(b) - want)) < TOL * 4
    c = np.random.default_rng(14).uniform(-300, 300, size=(200, 4, 4)).astype(np.float32)
    back = sfft.idctn(c.astype(np.float64), type=2, norm="ortho", axes=(-2, -1))
    assert np.max(np.abs(orc.idct4x4(c) - back)) < TOL


def test_a_transposed_dct_would_fail_this_file():
    """The check the older property tests could not make: a transform with rows and columns swapped is still orthonormal."""
    b = _blocks(20, 8, 15)
    want = sfft.dctn(b.astype(np.float64), type=2, norm="ortho", axes=(-2, -1))
    transposed = np.swapaxes(orc.dct8x8(b), -1, -2)
    assert np.max(np.abs(transposed[:, 1:, 1:] - want[:, 1:, 1:])) > 1.0


@pytest.mark.parametrize("v,h", [(2, 1), (1, 2), (0, 3), (5, 0)])
def test_basis_image_lands_at_its_own_coefficient(v, h):
    """Basis image of vertical frequency v (along rows, axis 0) and horizontal frequency h: cos((2r+1)v pi/16) cos((2x+1)h pi/16).
    Its only non-zero coefficient is [v][h] -- the reference modulates [2][1] (dct_encoder.py:33-35) and the kernels add
    d * c2[r] * c1[x] (csrc/common.hiph: c2_of / c1_of), so the index ORDER is what this pins."""
    r = np.arange(8)
    cv = np.cos((2 * r + 1) * v * np.pi / 16) * (np.sqrt(1 / 8) if v == 0 else 0.5)
    ch = np.cos((2 * r + 1) * h * np.pi / 16) * (np.sqrt(1 / 8) if h == 0 else 0.5)
    img = (100.0 * np.outer(cv, ch)).astype(np.float32)
    c = orc.dct8x8(img[None])[0]
    assert abs(c[v, h] - 100.0) < 1e-3
    rest = c.copy()
    rest[v, h] = 0
    assert np.max(np.abs(rest)) < 1e-4
    # and the inverse puts a lone coefficient back as that basis image
    e = np.zeros((8, 8), np.float32)
    e[v, h] = 100.0
    assert np.max(np.abs(orc.idct8x8(e[None])[0] - img)) < 1e-4


def test_kernel_rank1_constants_are_the_2_1_basis():
    """c2_of(r) x c1_of(x) in csrc/common.hiph is idct(e21): checked against scipy, not against the oracle."""
    e = np.zeros((8, 8))
    e[2, 1] = 1.0
    basis = sfft.idctn(e, type=2, norm="ortho")
    r = np.arange(8)
    c2 = 0.5 * np.cos((2 * r + 1) * 2 * np.pi / 16)
    c1 = 0.5 * np.cos((2 * r + 1) * 1 * np.pi / 16)
    assert np.max(np.abs(np.outer(c2, c1) - basis)) < 1e-12


def test_haar_dwt2_closed_form_and_subband_order():
    """pywt.dwt2(x, 'haar') -> (cA, (cH, cV, cD)), PyWavelets' documented convention: cH = detail along axis 0 (rows), approximation
    along axis 1 ('da'); cV = 'ad'; cD = 'dd'; haar's dec_hi = [-1, 1]/sqrt(2) under pywt's convolution gives even - odd.  With
    a, b / c, d the 2x2 cell (a b on the even row):  cA = (a+b+c+d)/2, cH = (a+b-c-d)/2, cV = (a-b+c-d)/2, cD = (a-b-c+d)/2."""
    x = np.random.default_rng(16).uniform(0, 255, size=(16, 24)).astype(np.float32)
    a, b, c, d = (x[0::2, 0::2].astype(np.float64), x[0::2, 1::2].astype(np.float64),
                  x[1::2, 0::2].astype(np.float64), x[1::2, 1::2].astype(np.float64))
    ca, (ch, cv, cd) = orc.haar_dwt2(x)
    assert ca.shape == (8, 12)
    assert np.max(np.abs(ca - (a + b + c + d) / 2)) < TOL
    assert np.max(np.abs(ch - (a + b - c - d) / 2)) < TOL
    assert np.max(np.abs(cv - (a - b + c - d) / 2)) < TOL
    assert np.max(np.abs(cd - (a - b - c + d) / 2)) < TOL
    # a purely vertical edge pattern (columns alternate) excites cV only; a horizontal one (rows alternate) cH only
    cols = np.tile(np.array([10.0, 30.0], np.float32), (8, 4))
    _, (h1, v1, d1) = orc.haar_dwt2(cols)
    assert np.all(h1 == 0) and np.all(d1 == 0) and np.all(np.abs(v1 + 20.0) < 1e-4)
    _, (h2, v2, d2) = orc.haar_dwt2(cols.T.copy())
    assert np.all(v2 == 0) and np.all(d2 == 0) and np.all(np.abs(h2 + 20.0) < 1e-4)
    assert np.max(np.abs(orc.haar_idwt2((ca, (ch, cv, cd))) - x)) < TOL
