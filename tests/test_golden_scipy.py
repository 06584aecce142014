"""Second fixture set: the reference's own modules run over an INDEPENDENT ``cv2`` / ``pywt`` stand-in.

tests/golden/ was captured with tools/standins/ handing the reference the ORACLE's dct / idct / cvtColor / dwt2 / idwt2,
which makes those vectors circular for the third-party primitives.  tests/golden_scipy/ was captured
by ``python tools/make_golden.py --standin scipy``: the same unmodified reference modules (dct_encoder.py:18-102,
dct_decoder.py:10-27, video/embedder.py:33-39, dwt_dct_svd_encoder.py:19-45, dwt_dct_svd_decoder.py:12-37) over
``scipy.fft.dctn / idctn(norm="ortho")`` on float32, ``cvtColor`` written from OpenCV's documented formula in plain NumPy
float32 and the closed-form 2x2 Haar (tools/standins_scipy/: nothing from the oracle).  The two fixture sets differ by
float rounding only, so they are compared at this repository's stated budgets, not bit for bit:

  payload after DeShuffler / DeGrayScale ... equal
  raw per-block bits ........................ <= 1e-4 of the blocks (floor: 1 block)
  marked u8 pixels .......................... <= 1 LSB on <= 1e-5 of the samples (floor: 1; DwtDctSvd: 2e-5) over
                                              sign-determined (DCT) / determined (DwtDctSvd) blocks; the number of
                                              blocks left out is printed per case (-s / -rP shows it)

CPU tests: the oracle against these vectors.  ``-m gpu`` tests: the HIP path (through the C ABI) against them.
What stays unpinned: OpenCV's and PyWavelets' own float rounding (neither library exists here or on the GPU box)."""
import os
import warnings

import numpy as np
import pytest

import offmark_oracle as orc
from conftest import ROOT

GOLDEN_SCIPY = os.path.join(ROOT, "tests", "golden_scipy")
C21_TOL = 1e-3
P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])


def dct_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_SCIPY) if f.endswith(".npz") and not f.startswith("svd_"))


def svd_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_SCIPY) if f.startswith("svd_") and f.endswith(".npz"))


def budget(n, frac, floor=1):
    return max(floor, int(np.floor(n * frac)))


def sign_determined(frame, wm, alpha):
    """(H, W) pixel mask and per-block mask of blocks whose |C21| exceeds the coefficient tolerance in the oracle."""
    enc = orc.DctEncoderOracle(alpha=alpha, promotion="nep50")
    enc.read_wm(wm)
    enc.encode(orc.bgr2yuv_f32(frame.astype(np.float32)))
    ok = np.abs(enc.debug["c21_pre"]) > C21_TOL
    H, W, _ = frame.shape
    m = np.ones((H, W), bool)
    m[: ok.shape[0] * 8, : ok.shape[1] * 8] = np.kron(ok, np.ones((8, 8), bool))
    return m, ok


def svd_determined(frame, wm, scales, blk=4):
    enc = orc.DwtDctSvdEncoderOracle(scales=scales, blk=blk)
    enc.read_wm(wm)
    enc.encode(orc.bgr2yuv_f32(frame.astype(np.float32)))
    ok = None
    for ch, dbg in enc.debug_ch.items():
        scale = float(scales[ch])
        s0, gap = dbg["s0"].astype(np.float64), dbg["gap"]
        frac = np.mod(s0, scale)
        this = (np.minimum(frac, scale - frac) > 1e-3 * np.maximum(1.0, s0 / 100)) & (gap < 1 - 1e-3)
        ok = this if ok is None else ok & this
    H, W, _ = frame.shape
    m = np.ones((H, W), bool)
    px = 2 * blk
    if ok is not None:
        m[: ok.shape[0] * px, : ok.shape[1] * px] = np.kron(ok, np.ones((px, px), bool))
    return m, ok


def check_pixels(got, ref, mask, frac, what):
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))[mask]
    if d.size == 0:
        return 0
    assert d.max() <= 1, f"{what}: max pixel diff {d.max()}"
    assert (d > 0).sum() <= budget(d.size, frac), f"{what}: {(d > 0).sum()} of {d.size} samples differ"
    return int((d > 0).sum())


def degenerate(g, raw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            if "image_payload" in g.files and bool(g["image_payload"]):
                return orc.degrayscale(raw, g["payload"].shape, int(g["key"]))
            return orc.deshuffle(raw, int(np.prod(g["payload"].shape)), int(g["key"]))


# ------------------------------------------------------------------------------------------------------------------
# CPU: the oracle against the independent-primitive vectors
# ------------------------------------------------------------------------------------------------------------------
def test_the_second_fixture_set_is_complete():
    assert len(dct_cases()) == 16 and len(svd_cases()) == 13          # the 16 DCT cases and the blk-4 DwtDctSvd cases


@pytest.mark.parametrize("case", dct_cases())
def test_oracle_dct_against_independent_primitive_vectors(case):
    g = np.load(os.path.join(GOLDEN_SCIPY, case + ".npz"))
    frame, alpha = g["frame"], float(g["alpha"])
    H, W, _ = frame.shape
    nblk = (H // 8) * (W // 8)
    L = int(np.prod(g["payload"].shape))
    enc = orc.DctEncoderOracle(alpha=alpha, promotion="nep50")
    enc.read_wm(g["wm"])
    marked = orc.mark_frame(frame, enc)
    mask, ok = sign_determined(frame, g["wm"], alpha)
    n_px = check_pixels(marked, g["marked"], mask, 1e-5, case)
    # the detector on the VECTOR's marked frame (isolates the read-out) and on the oracle's own marked frame
    dec = orc.DctDecoderOracle(alpha=alpha, promotion="nep50")
    raw = orc.check_frame(g["marked"], dec)
    assert raw.shape == g["raw_bits"].shape
    n_bits = int((raw != g["raw_bits"]).sum())
    assert n_bits <= budget(nblk, 1e-4), f"{n_bits} raw bits of {nblk} differ"
    own = orc.check_frame(marked, dec).reshape(-1)[:nblk][ok.reshape(-1)]
    n_own = int((own != g["raw_bits"].reshape(-1)[:nblk][ok.reshape(-1)]).sum())
    assert n_own <= budget(nblk, 1e-4)
    assert np.array_equal(np.asarray(degenerate(g, raw)).reshape(-1), np.asarray(g["degenerated"]).reshape(-1))
    # the masks, where the vectors hold them: float64 values carrying the float32 rounding of two different DCTs
    assert np.max(np.abs(enc.debug["lum"] - g["lum_mask"])) <= 2e-6
    flips = np.abs(enc.debug["tex"] - g["tex_mask"]) > 2e-6
    assert flips.sum() <= budget(nblk, 1e-4, floor=0), f"{int(flips.sum())} texture-mask branch flips"
    print(f"{case}: {int((~ok).sum())} of {nblk} blocks sign-ambiguous; {n_px} determined-block samples 1 LSB off; "
          f"{n_bits} raw bits differ on the vector's marked frame, {n_own} on the oracle's own")


@pytest.mark.parametrize("case", svd_cases())
def test_oracle_svd_against_independent_primitive_vectors(case):
    g = np.load(os.path.join(GOLDEN_SCIPY, case + ".npz"))
    frame = g["frame"]
    H, W, _ = frame.shape
    scales = tuple(float(x) for x in g["scales"]) if "scales" in g.files else (0.0, 15.0, 0.0)
    nblk = ((H // 4 * 2) // 4) * ((W // 4 * 2) // 4)
    enc = orc.DwtDctSvdEncoderOracle(scales=scales)
    enc.read_wm(g["wm"])
    marked = orc.mark_frame(frame, enc)
    mask, ok = svd_determined(frame, g["wm"], scales)
    n_px = check_pixels(marked, g["marked"], mask, 2e-5, case)
    raw = orc.check_frame(g["marked"], orc.DwtDctSvdDecoderOracle(scales=scales))
    assert raw.shape == g["raw_bits"].shape
    n_bits = int((raw != g["raw_bits"]).sum())
    assert n_bits <= budget(nblk, 1e-4), f"{n_bits} raw bits of {nblk} differ"
    assert np.array_equal(degenerate(g, raw), g["degenerated"])
    left_out = 0 if ok is None else int((~ok).sum())
    print(f"{case}: {left_out} of {nblk} blocks not determined; {n_px} determined-block samples 1 LSB off; {n_bits} raw bits differ")


# ------------------------------------------------------------------------------------------------------------------
# GPU: the HIP path against the same vectors
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def eng():
    import torch
    from offmark.engine import DctEngine
    torch.cuda.set_device(0)
    return DctEngine()


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("case", dct_cases())
def test_hip_dct_against_independent_primitive_vectors(eng, case):
    from offmark.degenerator.de_grayscale import DeGrayScale
    from offmark.degenerator.de_shuffler import DeShuffler
    g = np.load(os.path.join(GOLDEN_SCIPY, case + ".npz"))
    frame, alpha = g["frame"], float(g["alpha"])
    H, W, _ = frame.shape
    N, nblk = H * W // 64, (H // 8) * (W // 8)
    L = int(np.prod(g["payload"].shape))
    marked = eng.embed(cuda(frame[None]), g["wm"], alpha=alpha)[0].cpu().numpy()
    mask, ok = sign_determined(frame, g["wm"], alpha)
    n_px = check_pixels(marked, g["marked"], mask, 1e-5, case)
    counts, bits = eng.detect(cuda(g["marked"][None]), L, alpha=alpha, want_bits=True)
    bits = bits[0].cpu().numpy()
    n_bits = int((bits.reshape(-1) != g["raw_bits"].reshape(-1)).sum())
    assert n_bits <= budget(nblk, 1e-4), f"{n_bits} raw bits of {nblk} differ"
    _, own = eng.detect(cuda(marked[None]), L, alpha=alpha, want_bits=True)
    det = ok.reshape(-1)
    n_own = int((own[0].cpu().numpy()[:nblk][det] != g["raw_bits"].reshape(-1)[:nblk][det]).sum())
    assert n_own <= budget(nblk, 1e-4)
    if nblk >= 4 * L:
        cls = DeGrayScale if bool(g["image_payload"]) else DeShuffler
        out = cls(key=int(g["key"])).set_shape(g["payload"].shape).degenerate_counts(counts[0].cpu().numpy(), N)
        assert np.array_equal(np.asarray(out).reshape(-1), np.asarray(g["degenerated"]).reshape(-1))
    print(f"{case}: {int((~ok).sum())} of {nblk} blocks sign-ambiguous; {n_px} samples 1 LSB off; {n_bits} / {n_own} raw bits differ")


@pytest.mark.gpu
@pytest.mark.parametrize("case", svd_cases())
def test_hip_svd_against_independent_primitive_vectors(eng, case):
    from offmark.degenerator.de_shuffler import DeShuffler
    g = np.load(os.path.join(GOLDEN_SCIPY, case + ".npz"))
    frame = g["frame"]
    H, W, _ = frame.shape
    scales = tuple(float(x) for x in g["scales"]) if "scales" in g.files else (0.0, 15.0, 0.0)
    N, nblk = H * W // 64, ((H // 4 * 2) // 4) * ((W // 4 * 2) // 4)
    marked = eng.svd_embed(cuda(frame[None]), g["wm"], scales=scales)[0].cpu().numpy()
    mask, ok = svd_determined(frame, g["wm"], scales)
    n_px = check_pixels(marked, g["marked"], mask, 2e-5, case)
    counts, bits = eng.svd_detect(cuda(g["marked"][None]), 8, want_bits=True, scales=scales)
    bits = bits[0].cpu().numpy()
    n_bits = int((bits != g["raw_bits"].reshape(-1)).sum())
    assert n_bits <= budget(nblk, 1e-4), f"{n_bits} raw bits of {nblk} differ"
    if scales[1] > 0:
        out = DeShuffler(key=int(g["key"])).set_shape((8,)).degenerate_counts(counts[0].cpu().numpy(), N)
        assert np.array_equal(out, g["degenerated"])
    print(f"{case}: {0 if ok is None else int((~ok).sum())} of {nblk} blocks not determined; {n_px} samples 1 LSB off; {n_bits} raw bits differ")
