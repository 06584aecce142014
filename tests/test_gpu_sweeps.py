"""Seeded GPU-vs-oracle sweeps over every codec (DCT, DwtDctSvd blk = 4 and blk = 8): frame sizes that are multiples of nothing,
random payload lengths / keys / strengths, and content at the edges of the u8 cube and of the masks' branch conditions.
Budgets: the codec files' (tests/test_gpu_parity.py, tests/test_gpu_svd.py).  The first sweep found round 2's float32 overflow in
the 4x4 solver's eigenvector (bright Y blocks)."""
import numpy as np
import pytest

import offmark_oracle as orc
from test_gpu_svd import P8, cuda, determined_pixels, eng  # noqa: F401  (eng: the module-scoped engine fixture)

pytestmark = pytest.mark.gpu


def test_random_small_frames_all_codecs_against_oracle(eng):
    """Seeded sweep over frame sizes that are not multiples of anything in particular (fringes of every width, unaligned
    rows, fewer tiles than payload positions), payload lengths, keys, alphas / scales: DCT, DwtDctSvd blk = 4 and blk = 8,
    each against the oracle -- marked pixels over determined blocks, the read-out of the oracle's marked frame, the
    degenerated payload, the untouched fringe."""
    import warnings
    from offmark.degenerator.de_shuffler import DeShuffler
    rng = np.random.default_rng(20260)
    for case in range(36):
        H, W = int(rng.integers(16, 97)), int(rng.integers(16, 121))
        L, key = int(rng.integers(1, 12)), int(rng.integers(0, 50))
        payload = rng.integers(0, 2, L)
        frame = orc.synthetic_frame(H, W, 7000 + case)
        wm = orc.shuffle_generate(payload, (1, H * W // 64), key)
        codec = ("dct", "svd4", "svd8")[case % 3]
        dev = cuda(frame[None])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                      # nan means of empty slices, as upstream
            if codec == "dct":
                alpha = float(rng.choice([10.0, 20.0, 35.5]))
                enc = orc.DctEncoderOracle(alpha=alpha)
                enc.read_wm(wm)
                ref = orc.mark_frame(frame, enc)
                got = eng.embed(dev, wm, alpha=alpha)[0].cpu().numpy()
                ok = np.abs(enc.debug["c21_pre"]) > 1e-3
                px, th, tw, nbits = 8, H // 8, W // 8, H * W // 64
                ref_bits = orc.check_frame(ref, orc.DctDecoderOracle(alpha=alpha)).reshape(-1)
                counts, bits = eng.detect(cuda(ref[None]), L, alpha=alpha, want_bits=True)
            else:
                blk = 4 if codec == "svd4" else 8
                scales = [(0, 15, 0), (0, 9.5, 0), (7, 15, 11)][case % 9 // 3]
                enc = orc.DwtDctSvdEncoderOracle(scales=scales, blk=blk)
                enc.read_wm(wm)
                ref = orc.mark_frame(frame, enc)
                got = eng.svd_embed(dev, wm, scales=scales, blk=blk)[0].cpu().numpy()
                px, th, tw, nbits = 2 * blk, (H // 4 * 2) // blk, (W // 4 * 2) // blk, H * W // 4 // (blk * blk)
                ok = determined_pixels(frame, wm, scales, blk)[1] if th * tw else np.zeros((0, 0), bool)
                ref_bits = orc.check_frame(ref, orc.DwtDctSvdDecoderOracle(scales=scales, blk=blk)).reshape(-1)
                counts, bits = eng.svd_detect(cuda(ref[None]), L, scales=scales, blk=blk, want_bits=True)
            mask = np.zeros((H, W), bool)
            mask[: th * px, : tw * px] = np.kron(ok, np.ones((px, px), bool))
            d = np.abs(got.astype(int) - ref.astype(int))
            assert d[mask].size == 0 or d[mask].max() <= 1, (case, codec, H, W, d[mask].max())
            assert (d[mask] > 0).sum() <= max(1, int(2e-5 * d[mask].size)), (case, codec, H, W)
            assert np.array_equal(got[th * px:], frame[th * px:]) and np.array_equal(got[:, tw * px:], frame[:, tw * px:]), (case, codec, H, W)
            b = bits[0].cpu().numpy()
            assert b.shape == ref_bits.shape == (nbits,) and (b != ref_bits).sum() <= 1, (case, codec, H, W, int((b != ref_bits).sum()))
            deg = DeShuffler(key=key).set_shape((L,))
            if (b != ref_bits).sum() == 0:
                assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), nbits), orc.deshuffle(ref_bits[None], L, key)), (case, codec)


def _extreme_frames(H, W, rng):
    """Content at the edges of the u8 cube and of the masks' branch conditions."""
    yy, xx = np.mgrid[0:H, 0:W]
    out = {
        "uniform_noise": rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
        "checkerboard_1px": np.repeat((((yy + xx) & 1) * 255).astype(np.uint8)[..., None], 3, axis=2),
        "stripes_2px": np.repeat(((((xx // 2) & 1)) * 255).astype(np.uint8)[..., None], 3, axis=2),
        "saturated_primaries": np.stack([((xx // 8 + yy // 8) % 3 == k) * 255 for k in range(3)], axis=2).astype(np.uint8),
        "dark_noise": rng.integers(0, 30, (H, W, 3), dtype=np.uint8),
        "bright_noise": rng.integers(225, 256, (H, W, 3), dtype=np.uint8),
        "blue_ramp": np.stack([np.clip(xx * 3, 0, 255), np.zeros_like(xx), np.clip(255 - yy * 3, 0, 255)], axis=2).astype(np.uint8),
        "sparse_impulses": (rng.random((H, W, 3)) > 0.98).astype(np.uint8) * 255,
    }
    return out


@pytest.mark.parametrize("codec", ["dct", "svd4", "svd8"])
def test_extreme_content_against_oracle(eng, codec):
    """Noise over the whole u8 range, one-pixel checkerboards (all the energy in the highest frequencies: eh >> 900,
    l / e and (l + e) / h at their extremes), saturated primaries (clipping in the inverse colour transform), near-black
    and near-white noise (the luminance mask's dark branches; bright Y blocks), ramps and sparse impulses -- every codec
    against the oracle: marked pixels over determined blocks, the read-out of the oracle's marked frame, the payload."""
    from offmark.degenerator.de_shuffler import DeShuffler
    rng = np.random.default_rng(31)
    H, W = 96, 128
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
    deg = DeShuffler(key=0).set_shape((8,))
    for name, frame in _extreme_frames(H, W, rng).items():
        dev = cuda(frame[None])
        if codec == "dct":
            enc = orc.DctEncoderOracle(alpha=20)
            enc.read_wm(wm)
            ref = orc.mark_frame(frame, enc)
            got = eng.embed(dev, wm)[0].cpu().numpy()
            ok, px, nbits = np.abs(enc.debug["c21_pre"]) > 1e-3, 8, H * W // 64
            dec_o = orc.DctDecoderOracle(alpha=20)
            ref_bits = orc.check_frame(ref, dec_o).reshape(-1)
            counts, bits = eng.detect(cuda(ref[None]), 8, want_bits=True)
            r = dec_o.debug["c21"].astype(np.float64) / (20.0 * dec_o.debug["mask"])
            readable = (np.abs(np.abs(r - np.floor(r)) - 0.5) > 1e-4).reshape(-1)
            planes = eng.debug_planes(dev[0], alpha=20)
            for k in ("lum", "tex"):                 # tests/test_gpu_parity.py's rule: masks within 2e-6 except on threshold-flip blocks
                flips = np.abs(planes[k] - enc.debug[k]) > 2e-6
                assert flips.sum() <= max(1, int(1e-4 * flips.size)), (name, k, int(flips.sum()))
        else:
            blk = 4 if codec == "svd4" else 8
            scales = (6, 15, 9)
            enc = orc.DwtDctSvdEncoderOracle(scales=scales, blk=blk)
            enc.read_wm(wm)
            ref = orc.mark_frame(frame, enc)
            got = eng.svd_embed(dev, wm, scales=scales, blk=blk)[0].cpu().numpy()
            _, ok = determined_pixels(frame, wm, scales, blk)
            px, nbits = 2 * blk, H * W // 4 // (blk * blk)
            dec_o = orc.DwtDctSvdDecoderOracle(scales=scales, blk=blk)
            ref_bits = orc.check_frame(ref, dec_o).reshape(-1)
            counts, bits = eng.svd_detect(cuda(ref[None]), 8, scales=scales, blk=blk, want_bits=True)
            s0 = dec_o.debug["s0"].astype(np.float64).reshape(-1)
            m = np.mod(s0, 15.0)
            readable = np.minimum(np.minimum(m, 15.0 - m), np.abs(m - 7.5)) > 2e-6 * s0 + 1e-5
        mask = np.kron(ok, np.ones((px, px), bool))
        d = np.abs(got.astype(int) - ref.astype(int))[: mask.shape[0], : mask.shape[1]][mask]
        assert d.size == 0 or d.max() <= 1, (name, codec, int(d.max()))
        assert (d > 0).sum() <= max(2, int(1e-3 * d.size)), (name, codec, int((d > 0).sum()), d.size)
        b = bits[0].cpu().numpy()
        nb = readable.size
        assert (b[:nb] != ref_bits[:nb])[readable].sum() <= 1, (name, codec, int((b[:nb] != ref_bits[:nb])[readable].sum()))
        if (b != ref_bits).sum() == 0:
            assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), nbits), orc.deshuffle(ref_bits[None], 8, 0)), (name, codec)
