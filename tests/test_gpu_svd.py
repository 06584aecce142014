"""GPU parity for the DwtDctSvd codec -- the pair tests/mark.py and tests/detect.py construct.

The svd_*.npz vectors were captured by running the reference's own modules with the oracle's restated pywt /
cv2 primitives supplied as stand-in modules (neither library is installed): they pin the reference's control
flow; PyWavelets' and OpenCV's float arithmetic is PARITY UNPINNED (np.linalg.svd is the real LAPACK).

Tolerances (none pinned upstream), ~10x what is measured (profiles/r2_parity_stats.txt: 2.1e-6 and 0):
  payload after DeShuffler ... bit-exact
  raw per-block bits ......... <= 1e-4 of the blocks (floor: 1 block) vs oracle
  marked u8 pixels ........... <= 1 LSB on <= 2e-5 of the samples (floor: 1) over "determined" blocks: a block is
     skipped when its top singular value is within 1e-3 of a multiple of the quantisation step
     (s0 // scale flips on the last float bits and moves s0 by a whole step) or when its two largest
     singular values are within 1e-3 relative (the rank-1 direction u0 v0^T is then not defined).
"""
import os

import numpy as np
import pytest

import offmark_oracle as orc
from conftest import GOLDEN, svd_golden_cases

pytestmark = pytest.mark.gpu
P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])


@pytest.fixture(scope="module")
def eng():
    import torch
    from offmark.engine import DctEngine
    torch.cuda.set_device(0)
    return DctEngine()


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def budget(n, frac, floor=1):
    return max(floor, int(np.floor(n * frac)))


def determined_pixels(frame, wm, scales=(0, 15, 0), blk=4):
    """Blocks whose marking is defined to float32 accuracy in EVERY marked channel (see the module text)."""
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
    px = 2 * blk                                     # pixels per tile side
    m[: ok.shape[0] * px, : ok.shape[1] * px] = np.kron(ok, np.ones((px, px), bool))
    return m, ok


def assert_pixels_close(got, ref, mask):
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))[mask]
    if d.size:
        assert d.max() <= 1, f"max pixel diff {d.max()}"
        assert (d > 0).sum() <= budget(d.size, 2e-5), f"{(d > 0).sum()} of {d.size} samples differ"


@pytest.mark.parametrize("case", svd_golden_cases())
def test_svd_golden_embed_and_detect(eng, case):
    from offmark.degenerator.de_shuffler import DeShuffler
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    frame = g["frame"]
    H, W, _ = frame.shape
    scales = tuple(float(x) for x in g["scales"]) if "scales" in g.files else (0.0, 15.0, 0.0)      # round 2: per-channel scales
    blk = int(g["blk"]) if "blk" in g.files else 4                                                  # round 3: blk = 8
    px = 2 * blk
    th, tw = (H // 4 * 2) // blk, (W // 4 * 2) // blk                 # tiles (dwt_dct_svd_encoder.py:29-32 on the LL band)
    N, nblk = H * W // 4 // (blk * blk), th * tw                       # the decoder's bit count (decoder.py:14) and the tiles it fills
    marked = eng.svd_embed(cuda(frame[None]), g["wm"], scales=scales, blk=blk)[0].cpu().numpy()
    mask, ok = determined_pixels(frame, g["wm"], scales, blk)
    if nblk >= 32:
        assert ok.mean() > (0.9 if sum(x > 0 for x in scales) == 1 else 0.8)
    assert_pixels_close(marked, g["marked"], mask)
    assert np.array_equal(marked[th * px:], frame[th * px:]) and np.array_equal(marked[:, tw * px:], frame[:, tw * px:])
    if not (scales[0] > 0 or scales[2] > 0):
        assert np.array_equal(marked[..., 2], frame[..., 2])          # channel 2 untouched when only U is marked
    counts, bits = eng.svd_detect(cuda(g["marked"][None]), 8, want_bits=True, scales=scales, blk=blk)
    bits = bits[0].cpu().numpy()
    assert bits.shape == (N,) == g["raw_bits"].reshape(-1).shape and not bits[nblk:].any()
    assert (bits != g["raw_bits"].reshape(-1)).sum() <= budget(nblk, 1e-4)
    assert np.array_equal(counts[0].cpu().numpy(), np.array([bits[i::8].sum() for i in range(8)]))
    if scales[1] > 0:
        out = DeShuffler(key=int(g["key"])).set_shape((8,)).degenerate_counts(counts[0].cpu().numpy(), N)
        assert np.array_equal(out, g["degenerated"])
    else:                                            # the reference reads channel 1 only (decoder.py:24): zeros
        assert not bits.any() and not g["raw_bits"].any()
    # fused embed+verify == embed followed by detect
    o2, c2, b2 = eng.svd_embed_detect(cuda(frame[None]), g["wm"], 8, want_bits=True, scales=scales, blk=blk)
    c3, b3 = eng.svd_detect(o2, 8, want_bits=True, scales=scales, blk=blk)
    import torch
    assert torch.equal(c2, c3) and torch.equal(b2, b3) and np.array_equal(o2[0].cpu().numpy(), marked)
    # in place, and the plugin classes (DwtDctSvdEncoder(blk=...).encode / DwtDctSvdDecoder(blk=...).decode on float32 YUV)
    buf = cuda(frame[None]).clone()
    eng.svd_embed(buf, g["wm"], scales=scales, blk=blk, out=buf)
    assert np.array_equal(buf[0].cpu().numpy(), marked)
    if blk != 4 and "yuv_in" in g.files:
        from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder
        from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder
        enc = DwtDctSvdEncoder(scales=list(scales), blk=blk)
        enc.read_wm(g["wm"])
        got = enc.encode(g["yuv_in"].copy())
        tile_ok = np.kron(ok, np.ones((px, px), bool))
        for ch in range(3):
            d = np.abs(got[: th * px, : tw * px, ch] - g["yuv_out"][: th * px, : tw * px, ch])
            assert d[tile_ok].max() <= 3e-3, ch
            if not scales[ch] > 0:
                assert np.array_equal(got[:, :, ch], g["yuv_in"][:, :, ch])
        dec = DwtDctSvdDecoder(scales=list(scales), blk=blk)
        rb = dec.decode(g["yuv_out"].copy())
        assert rb.shape == (1, N) and dec.block_num == N
        assert (rb != g["raw_bits_clean"]).reshape(-1)[: nblk][ok.reshape(-1)].sum() <= budget(nblk, 1e-4)


def test_svd_1080p_against_oracle_and_payloads(eng):
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    H, W = 1080, 1920
    frame = orc.synthetic_frame(H, W, 2000)
    wm = orc.shuffle_generate(P8, (1, 32400), 0)
    enc = orc.DwtDctSvdEncoderOracle()
    enc.read_wm(wm)
    ref = orc.mark_frame(frame, enc)
    marked, counts, bits = eng.svd_embed_detect(cuda(frame[None]), wm, 8, want_bits=True)
    mask, ok = determined_pixels(frame, wm)
    assert_pixels_close(marked[0].cpu().numpy(), ref, mask)
    ref_bits = orc.check_frame(ref, orc.DwtDctSvdDecoderOracle())
    _, b2 = eng.svd_detect(cuda(ref[None]), 8, want_bits=True)
    assert (b2[0].cpu().numpy() != ref_bits.reshape(-1)).sum() <= budget(32400, 1e-4)
    deg = DeShuffler(key=0).set_shape((8,))
    assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), 32400), P8)


def test_svd_blk8_full_frames_against_oracle_and_payloads(eng):
    """DwtDctSvd*(blk=8) at 1080p (67 x 120 tiles of 16x16 pixels, 4 LL rows of fringe) and on the reference's natural frame:
    marked pixels against the oracle over determined tiles, the read-out of the oracle's marked frame, payloads, the
    per-channel-scales form, several frames with own rows, and that nothing outside the tiles changes."""
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    from conftest import natural_frame
    H, W = 1080, 1920
    N8, th, tw = H * W // 256, (H // 4 * 2) // 8, (W // 4 * 2) // 8
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
    deg = DeShuffler(key=0).set_shape((8,))
    for frame, scales in ((orc.synthetic_frame(H, W, 2000), (0, 15, 0)), (natural_frame(), (0, 15, 0)), (orc.synthetic_frame(H, W, 2001), (9, 15, 21))):
        enc = orc.DwtDctSvdEncoderOracle(scales=scales, blk=8)
        enc.read_wm(wm)
        ref = orc.mark_frame(frame, enc)
        marked, counts, bits = eng.svd_embed_detect(cuda(frame[None]), wm, 8, want_bits=True, scales=scales, blk=8)
        mask, ok = determined_pixels(frame, wm, scales, blk=8)
        assert ok.shape == (th, tw) and ok.mean() > 0.8
        m = marked[0].cpu().numpy()
        assert_pixels_close(m, ref, mask)
        assert np.array_equal(m[th * 16:], frame[th * 16:]) and np.array_equal(m[:, tw * 16:], frame[:, tw * 16:])
        ref_bits = orc.check_frame(ref, orc.DwtDctSvdDecoderOracle(scales=scales, blk=8)).reshape(-1)
        assert ref_bits.shape == (N8,)
        _, b2 = eng.svd_detect(cuda(ref[None]), 8, want_bits=True, scales=scales, blk=8)
        assert (b2[0].cpu().numpy() != ref_bits).sum() <= budget(th * tw, 1e-4)
        assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), N8), P8)
    # a batch with per-frame rows (the clamp included) equals frame-by-frame calls
    frames = cuda(np.stack([orc.synthetic_frame(240, 320, 1001 + i) for i in range(5)]))
    table = np.stack([orc.shuffle_generate(np.roll(P8, i), (1, 1200), 0)[0] for i in range(3)]).astype(np.uint8)
    rows = [2, 0, 1, 1, 2]
    batch = eng.svd_embed(frames, table, wm_row=rows, blk=8)
    for i, r in enumerate(rows):
        assert torch.equal(batch[i], eng.svd_embed(frames[i:i + 1], table[r:r + 1], blk=8)[0])
    c, _ = eng.svd_detect(batch, 8, blk=8)
    got = deg.degenerate_counts(c.cpu().numpy(), 240 * 320 // 256)
    assert all(np.array_equal(got[i], np.roll(P8, rows[i])) for i in range(5))
    # frames smaller than one tile: nothing to mark, nothing to read
    tiny = cuda(orc.synthetic_frame(12, 40, 3)[None])
    assert torch.equal(eng.svd_embed(tiny, np.zeros((1, 7), np.uint8), blk=8), tiny)
    c0, b0 = eng.svd_detect(tiny, 8, want_bits=True, blk=8)
    assert not c0.any() and b0.shape == (1, 12 * 40 // 256) and not b0.any()


def test_mark_py_and_detect_py_logic_literally(eng):
    """tests/mark.py:18-40 and tests/detect.py:17-31 with the in-memory reader/writer in place of the
    ffmpeg pipes: same imports, same calls, same payload."""
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder
    from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder
    from offmark.generator.shuffler import Shuffler
    from offmark.video.embedder import Embedder
    from offmark.video.extractor import Extractor
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark.video.frame_writer import ArrayFrameWriter
    frames = np.stack([orc.synthetic_frame(240, 320, 1001 + i) for i in range(24)])
    payload = np.array([0, 1, 1, 0, 0, 1, 0, 1])
    r = ArrayFrameReader(frames)
    w = ArrayFrameWriter()
    frame_embedder = DwtDctSvdEncoder()
    capacity = frame_embedder.wm_capacity((r.height, r.width, 3))
    generator = Shuffler(key=0)
    wm = generator.generate_wm(payload, capacity)
    frame_embedder.read_wm(wm)
    Embedder(r, frame_embedder, w).start()
    degenerator = DeShuffler(key=0)
    degenerator.set_shape(payload.shape)
    video_extractor = Extractor(ArrayFrameReader(w.frames), DwtDctSvdDecoder(), degenerator)
    video_extractor.start()
    assert len(video_extractor.patterns) == 24 and all(np.array_equal(p, payload) for p in video_extractor.patterns)
    # plugin-level float32 YUV boundary against the oracle
    yuv = orc.bgr2yuv_f32(frames[0].astype(np.float32))
    ref_enc = orc.DwtDctSvdEncoderOracle()
    ref_enc.read_wm(wm)
    ref = ref_enc.encode(yuv.copy())
    arg = yuv.copy()
    got = frame_embedder.encode(arg)
    assert got is arg and np.array_equal(got[:, :, 0], yuv[:, :, 0]) and np.array_equal(got[:, :, 2], yuv[:, :, 2])
    _, ok = determined_pixels(frames[0], wm)
    blk_ok = np.kron(ok, np.ones((8, 8), bool))
    assert np.abs(got[:, :, 1] - ref[:, :, 1])[blk_ok].max() <= 2e-3
    bits = DwtDctSvdDecoder().decode(ref)
    ref_bits = orc.DwtDctSvdDecoderOracle().decode(ref)
    assert bits.dtype == np.float64 and bits.shape == ref_bits.shape
    # raw bits at the float32 plugin boundary: the file's budget (1e-4 of the blocks, floor 1) over determined blocks --
    # the others (s0 within 1e-3 of a multiple of the step, or s1 ~ s0: the marking itself is not defined to float32
    # accuracy there) are masked as in the u8 tests
    assert (bits != ref_bits).reshape(-1)[ok.reshape(-1)].sum() <= budget(int(ok.sum()), 1e-4)
    with pytest.raises(NotImplementedError):
        DwtDctSvdEncoder(blk=2)                          # indexes past the reference's own watermark; not built
    assert DwtDctSvdEncoder(blk=8).blk == 8              # round 3: 16x16 pixel tiles (golden cases svd_blk8_*)
    with pytest.raises(ValueError):
        DwtDctSvdEncoder(scales=[0, 0, 0])
    with pytest.raises(ValueError):
        DwtDctSvdEncoder(scales=[0, 1e-46, 0])          # a positive scale that float32 flushes to zero marks nothing
    # per-channel scales at the plugin boundary (dwt_dct_svd_encoder.py:19-26): every marked channel changes
    sc = [10, 15, 20]
    enc3 = DwtDctSvdEncoder(scales=sc)
    enc3.read_wm(wm)
    ref3 = orc.DwtDctSvdEncoderOracle(scales=sc)
    ref3.read_wm(wm)
    want = ref3.encode(yuv.copy())
    got3 = enc3.encode(yuv.copy())
    _, ok3 = determined_pixels(frames[0], wm, sc)
    blk3 = np.kron(ok3, np.ones((8, 8), bool))
    for ch in range(3):
        assert np.abs(got3[:, :, ch] - want[:, :, ch])[blk3].max() <= 3e-3, ch
        assert not np.array_equal(got3[:, :, ch], yuv[:, :, ch])
    bits3 = DwtDctSvdDecoder(scales=sc).decode(want)
    assert (bits3 != orc.DwtDctSvdDecoderOracle(scales=sc).decode(want)).reshape(-1)[ok3.reshape(-1)].sum() <= budget(int(ok3.sum()), 1e-4)
    assert not DwtDctSvdDecoder(scales=[12, 0, 0]).decode(want).any()


@pytest.mark.parametrize("scales", [(10.0, 15.0, 20.0), (12.0, 0.0, 0.0), (0.0, 0.0, 30.0), (7.5, 22.0, 0.0)])
def test_svd_per_channel_scales_against_oracle(eng, scales):
    """DwtDctSvdEncoder(scales=[a, b, c]) (dwt_dct_svd_encoder.py:6,19-26): any subset of the YUV channels marked,
    each with its own step; the read-out stays channel 1's (dwt_dct_svd_decoder.py:24)."""
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    for (H, W, seed) in [(240, 320, 1001), (1080, 1920, 2000), (36, 52, 6)]:
        N, nblk = H * W // 64, (H // 8) * (W // 8)
        frame = orc.synthetic_frame(H, W, seed)
        wm = orc.shuffle_generate(P8, (1, N), 0)
        enc = orc.DwtDctSvdEncoderOracle(scales=scales)
        enc.read_wm(wm)
        ref = orc.mark_frame(frame, enc)
        marked, counts, bits = eng.svd_embed_detect(cuda(frame[None]), wm, 8, want_bits=True, scales=scales)
        mask, ok = determined_pixels(frame, wm, scales)
        assert_pixels_close(marked[0].cpu().numpy(), ref, mask)
        ref_bits = orc.check_frame(ref, orc.DwtDctSvdDecoderOracle(scales=scales)).reshape(-1)
        c2, b2 = eng.svd_detect(cuda(ref[None]), 8, want_bits=True, scales=scales)
        assert (b2[0].cpu().numpy()[:ref_bits.size] != ref_bits).sum() <= budget(nblk, 1e-4)
        c3, b3 = eng.svd_detect(marked, 8, want_bits=True, scales=scales)
        assert torch.equal(c3, counts) and torch.equal(b3, bits)
        if scales[1] > 0 and nblk >= 32:
            assert np.array_equal(DeShuffler(key=0).set_shape((8,)).degenerate_counts(counts[0].cpu().numpy(), N), P8)
        if not scales[1] > 0:
            assert not bits.any() and not ref_bits.any()


def test_svd_random_shapes_and_contents(eng):
    """Seeded sweep over frame sizes, batch sizes and contents, every frame against the oracle."""
    from test_gpu_parity import _content
    rng = np.random.default_rng(777)
    kinds = ["synthetic", "noise", "dark", "bright", "ramp", "checker", "flat"]
    for trial in range(20):
        H, W = int(rng.integers(8, 150)), int(rng.integers(8, 200))
        if trial % 4 == 0:
            W = (W // 16 + 1) * 16
        n = int(rng.integers(1, 4))
        N, nblk = H * W // 64, (H // 8) * (W // 8)
        wm = orc.shuffle_generate(rng.integers(0, 2, 8), (1, N), 1)
        frames = np.stack([_content(rng, H, W, kinds[(trial + k) % len(kinds)]) for k in range(n)])
        got, counts, bits = eng.svd_embed_detect(cuda(frames), wm, 8, want_bits=True)
        got = got.cpu().numpy()
        for k in range(n):
            enc = orc.DwtDctSvdEncoderOracle()
            enc.read_wm(wm)
            ref = orc.mark_frame(frames[k], enc)
            mask, _ = determined_pixels(frames[k], wm)
            assert_pixels_close(got[k], ref, mask)
            ref_bits = orc.check_frame(ref, orc.DwtDctSvdDecoderOracle())
            _, b2 = eng.svd_detect(cuda(ref[None]), 8, want_bits=True)
            assert (b2[0].cpu().numpy() != ref_bits.reshape(-1)).sum() <= budget(nblk, 1e-4), (trial, k)
        c3, b3 = eng.svd_detect(cuda(got), 8, want_bits=True)
        assert np.array_equal(c3.cpu().numpy(), counts.cpu().numpy()) and np.array_equal(b3.cpu().numpy(), bits.cpu().numpy())


def test_non_finite_and_extreme_tiles_stay_local_and_terminate(eng):
    """The DwtDctSvd codec has no frame-global step, so a tile of nan / inf / 1e30 / denormals in a float32 YUV frame (the
    plugin boundary takes any floats) must neither stall the solver's wave-uniform loop (it is capped) nor change any
    other tile: every clean tile comes out bit-identical to the same frame without the poison, and the read-out of the
    clean tiles is unchanged.  (What the poisoned tiles themselves become is not specified -- upstream LAPACK raises or
    returns nan there.)"""
    import torch
    rng = np.random.default_rng(31)
    H, W = 64, 128
    rgb = rng.integers(0, 256, (H, W, 3)).astype(np.float32)
    rgb[:, :64] = np.linspace(20, 230, 64, dtype=np.float32)[None, :, None]            # a smooth half and a noisy half
    clean = orc.bgr2yuv_f32(rgb)
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
    poisons = [np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-40, 3e38]
    bad = clean.copy()
    where = []
    for k, v in enumerate(poisons):
        bi, bj = k % (H // 8), (3 * k + 1) % (W // 8)
        bad[bi * 8:bi * 8 + 8, bj * 8:bj * 8 + 8, 1] = v
        if k % 2:
            bad[bi * 8 + 3, bj * 8 + 5, 1] = 7.0                                       # a mixed tile
        where.append((bi, bj))
    for scales in (None, (9.0, 15.0, 20.0)):
        out_clean = eng.svd_encode_yuv(cuda(clean[None].copy()), wm, scales=scales)
        out_bad = eng.svd_encode_yuv(cuda(bad[None].copy()), wm, scales=scales)
        torch.cuda.synchronize()                                                      # returns: the loop terminated
        a, b = out_clean[0].cpu().numpy(), out_bad[0].cpu().numpy()
        ok = np.ones((H // 8, W // 8), bool)
        for bi, bj in where:
            ok[bi, bj] = False
        m = np.kron(ok, np.ones((8, 8), bool))
        assert np.array_equal(a[m], b[m])
        bits_clean = eng.svd_decode_yuv(out_clean, scales=scales)[0].cpu().numpy()[: ok.size].reshape(ok.shape)
        bits_bad = eng.svd_decode_yuv(out_bad, scales=scales)[0].cpu().numpy()[: ok.size].reshape(ok.shape)
        assert np.array_equal(bits_clean[ok], bits_bad[ok])
        assert np.array_equal(bits_clean[ok], np.asarray(wm).reshape(-1)[: ok.size].reshape(ok.shape)[ok].astype(np.uint8))


@pytest.mark.parametrize("blk", [4, 8])
def test_bright_frames_with_the_luma_channel_marked(eng, blk):
    """Regression (round 3): LL blocks of a bright Y channel have s0 up to 2040; in the 4x4 solver the squared norm of the
    adjugate column (cofactors ~ s0^6) overflowed float32, the vector came out as zero and the block was marked like a zero
    block.  Near-white, white and mid-grey content with scales[0] > 0, both block sizes, against the oracle."""
    rng = np.random.default_rng(77)
    H, W = 64, 96
    frames = [np.clip(rng.normal(238, 12, (H, W, 3)), 0, 255).astype(np.uint8), np.full((H, W, 3), 255, np.uint8),
              np.clip(rng.normal(128, 40, (H, W, 3)), 0, 255).astype(np.uint8)]
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
    for frame in frames:
        for scales in ((7, 0, 0), (7, 15, 11), (30, 0, 0)):
            enc = orc.DwtDctSvdEncoderOracle(scales=scales, blk=blk)
            enc.read_wm(wm)
            ref = orc.mark_frame(frame, enc)
            got = eng.svd_embed(cuda(frame[None]), wm, scales=scales, blk=blk)[0].cpu().numpy()
            mask, ok = determined_pixels(frame, wm, scales, blk)
            d = np.abs(got.astype(int) - ref.astype(int))[mask]
            # 1 LSB on <= 1e-3 of the samples: float32 carries s0 = 2000 to ~4e-4 absolute (LAPACK's float32 too), s0' - s0 and with
            # it every pixel's change inherit that, so ~2e-4 of the roundings land on the other side (U / V blocks, s0 <= 900 and
            # a smaller share of s0 in the pixel, hold the file's 2e-5)
            assert d.size == 0 or (d.max() <= 1 and (d > 0).sum() <= max(2, int(1e-3 * d.size))), (scales, d.max(), int((d > 0).sum()))
            if frame.mean() > 200:
                assert enc.debug_ch[0]["s0"].max() > 1700                                # the range whose adjugate norm overflowed (blk = 4)


@pytest.mark.parametrize("blk", [4, 8])
def test_svd_counts_for_long_payloads_use_the_global_atomic_path(eng, blk):
    """Payload lengths beyond the LDS histogram (2048): counts[i] must still be the number of ones among bits[i::L]
    (de_shuffler.py:17-18), also when L exceeds the number of bits a frame carries."""
    frames = cuda(np.stack([orc.synthetic_frame(240, 320, 1001 + i) for i in range(3)]))
    wm = cuda(np.random.default_rng(4).integers(0, 2, (1, 1200), dtype=np.uint8))
    marked = eng.svd_embed(frames, wm, blk=blk)
    for L in (8, 2048, 2049, 3000, 70000):
        counts, bits = eng.svd_detect(marked, L, want_bits=True, blk=blk)
        b = bits.cpu().numpy()
        want = np.stack([[row[i::L].sum() for i in range(L)] for row in b]) if L <= 3000 else None
        c = counts.cpu().numpy()
        if want is not None:
            assert np.array_equal(c, want), (blk, L)
        else:
            assert np.array_equal(c[:, : b.shape[1]], b) and not c[:, b.shape[1]:].any()       # every bit is alone in its position
        o, c2, b2 = eng.svd_embed_detect(frames, wm, L, want_bits=True, blk=blk)
        assert np.array_equal(c2.cpu().numpy(), c) and np.array_equal(b2.cpu().numpy(), b)
