"""Property tests (hypothesis) of the host-side codecs against the oracle: random payload lengths, keys,
capacities and bit-error patterns.  CPU only."""
import warnings

import numpy as np
from hypothesis import given, settings, strategies as st

import offmark_oracle as orc
from offmark.degenerator.de_grayscale import DeGrayScale
from offmark.degenerator.de_shuffler import DeShuffler
from offmark.engine import payload_means
from offmark.generator.grayscale import GrayScale
from offmark.generator.shuffler import Shuffler

keys = st.integers(min_value=0, max_value=2**31 - 1)


@settings(max_examples=150, deadline=None)
@given(L=st.integers(1, 40), cap=st.integers(1, 700), key=keys, seed=st.integers(0, 10**6))
def test_shuffler_roundtrip_and_oracle_agreement(L, cap, key, seed):
    rng = np.random.default_rng(seed)
    payload = rng.integers(0, 2, size=L)
    wm = Shuffler(key=key).generate_wm(payload, (1, cap))
    assert wm.shape == (1, cap) and np.array_equal(wm, orc.shuffle_generate(payload, (1, cap), key))
    noisy = wm.astype(np.float64).reshape(-1).copy()
    flips = rng.random(cap) < 0.15
    noisy[flips] = 1 - noisy[flips]
    deg = DeShuffler(key=key).set_shape(payload.shape)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.deshuffle(noisy, L, key)
        got = deg.degenerate(noisy)
    assert np.array_equal(got, ref)
    counts = np.array([noisy[i::L].sum() for i in range(L)]).astype(np.int64)
    assert np.array_equal(deg.degenerate_counts(counts, cap), ref)
    if cap >= 40 * L and 0 < payload.sum() < L and not flips.any():
        assert np.array_equal(got, payload)            # clean, non-constant payload survives the round trip


@settings(max_examples=60, deadline=None)
@given(h=st.integers(1, 9), w=st.integers(1, 9), cap=st.integers(1, 500), key=keys, seed=st.integers(0, 10**6))
def test_grayscale_codecs_match_oracle(h, w, cap, key, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wm = GrayScale(key=key).generate_wm(img, (1, cap))
        assert np.array_equal(wm, orc.grayscale_generate(img, (1, cap), key))
        bits = wm.astype(np.float64)
        got = DeGrayScale(key=key).set_shape(img.shape).degenerate(bits)
        assert got.shape == img.shape and np.array_equal(got, orc.degrayscale(bits, img.shape, key))


@settings(max_examples=100, deadline=None)
@given(L=st.integers(1, 64), N=st.integers(0, 2000))
def test_payload_means_slice_lengths(L, N):
    bits = np.ones(N)
    counts = np.array([bits[i::L].sum() for i in range(L)])
    with np.errstate(all="ignore"):
        means = payload_means(counts, N, L)
    for i in range(L):
        if len(bits[i::L]):
            assert means[i] == 1.0
        else:
            assert np.isnan(means[i])


# ---- C oracle == NumPy oracle on random small frames (sizes need not be multiples of 8) ------------------
import c_oracle  # noqa: E402


@settings(max_examples=40, deadline=None)
@given(h=st.integers(8, 40), w=st.integers(8, 56), seed=st.integers(0, 10**6), alpha=st.sampled_from([5, 10, 20, 33.5]),
       legacy=st.booleans(), flat=st.booleans())
def test_c_oracle_matches_numpy_oracle_on_random_frames(h, w, seed, alpha, legacy, flat):
    rng = np.random.default_rng(seed)
    if flat:                                  # piecewise-constant content: exact-zero C21 and mask edge branches
        frame = np.kron(rng.integers(0, 256, size=((h + 7) // 8, (w + 7) // 8, 3)), np.ones((8, 8, 1)))[:h, :w].astype(np.uint8)
    else:
        frame = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    n = h * w // 64
    wm = rng.integers(0, 2, size=(1, max(n, 1)))
    promo = "legacy" if legacy else "nep50"
    enc = orc.DctEncoderOracle(alpha=alpha, promotion=promo)
    enc.read_wm(wm)
    ref = orc.mark_frame(frame, enc)
    got, _ = c_oracle.mark_frames(frame[None], wm, alpha=alpha, legacy=legacy)
    assert np.array_equal(got[0], ref)
    bits_ref = orc.check_frame(ref, orc.DctDecoderOracle(alpha=alpha, promotion=promo)).reshape(-1)
    bits, _ = c_oracle.check_frames(ref[None], alpha=alpha, legacy=legacy)
    assert np.array_equal(bits[0], bits_ref)


def test_build_defined_yuv420_conversion_properties():
    """oracle.yuv420_to_rgb / rgb_to_yuv420 (the conversion is the build's own, not swscale's):
    studio-swing fixed points, layouts, and stability of the 4:2:0 round trip."""
    rng = np.random.default_rng(0)
    for g, y in ((0, 16), (255, 235), (128, 126)):                       # black, white, mid grey
        yy, u, v = orc.rgb_to_yuv420(np.full((8, 8, 3), g, np.uint8))
        assert (yy == y).all() and (u == 128).all() and (v == 128).all()
        assert (np.abs(orc.yuv420_to_rgb(yy, u, v).astype(int) - g) <= 1).all()
    red = np.zeros((8, 8, 3), np.uint8)
    red[..., 0] = 255
    yy, u, v = orc.rgb_to_yuv420(red)
    assert yy[0, 0] == 81 and u[0, 0] == 90 and v[0, 0] == 240          # BT.601 studio-swing red
    rgb = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    planes = orc.rgb_to_yuv420(rgb)
    for layout in ("i420", "nv12"):
        buf = orc.pack_yuv420(*planes, layout)
        assert buf.size == 32 * 48 * 3 // 2
        assert all(np.array_equal(a, b) for a, b in zip(orc.unpack_yuv420(buf, 32, 48, layout), planes))
    # chroma-flat content survives the round trip to within the two roundings; a second round trip changes little
    flat = np.broadcast_to(rng.integers(30, 220, 3, dtype=np.uint8), (16, 16, 3)).copy()
    once = orc.yuv420_to_rgb(*orc.rgb_to_yuv420(flat))
    twice = orc.yuv420_to_rgb(*orc.rgb_to_yuv420(once))
    assert np.abs(once.astype(int) - flat).max() <= 2 and np.abs(twice.astype(int) - once).max() <= 1
    # every input byte combination stays in range (clip) -- including illegal studio-swing values
    y, u, v = (rng.integers(0, 256, s, dtype=np.uint8) for s in ((16, 16), (8, 8), (8, 8)))
    out = orc.yuv420_to_rgb(y, u, v)
    assert out.dtype == np.uint8 and out.shape == (16, 16, 3)
