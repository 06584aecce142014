"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle and the golden vectors.

What is compared with what: the oracle (oracle/offmark_oracle.py) reproduces bit for bit the vectors in
tests/golden/, which were captured by running the reference's own Python modules with the oracle's restated
cv2.dct / cv2.idct / cvtColor supplied as a stand-in module (OpenCV is not installed: tools/make_golden.py).
So these tests pin the reference's control flow and scalar semantics; OpenCV's own float rounding is
PARITY UNPINNED.

Tolerances (the reference pins none for this path).  Budgets are ~10x what is measured
(profiles/r2_parity_stats.txt; every comparison made by this file is also logged, see _MEASURED):
  payload after DeShuffler ............ bit-exact
  raw per-block bits .................. <= 1e-4 of the blocks (floor: 1 block) vs oracle (ulp-level threshold flips)
  Y DC, C21 ........................... <= 1e-3 absolute on 0..255-scale data
  luminance / texture masks ........... within 2e-6 absolute (float32 rounding of the block mean / of eh carried through
                                        the float64 formulas) except on <= 1e-4 threshold-flip blocks
  marked u8 pixels .................... <= 1 LSB, on <= 1e-5 of the samples (floor: 1 sample), over
                                        "sign-determined" blocks

Sign-ambiguous blocks: the reference multiplies the quantised magnitude by np.sign(C21)
(dct_encoder.py:33-35).  Where |C21| is below the coefficient tolerance (1e-3) -- typical for
chroma-flat 8x8 blocks of JPEG/H.264-decoded video, where C21 is ~1e-6 of float rounding noise --
the sign is decided by the last bits of cv2.dct / cvtColor and no independent implementation can
reproduce it.  For those blocks the tests require the same quantised MAGNITUDE (so the decoded bit
is identical) and leave the sign free; pixels are compared on the remaining blocks.
"""
import os

import numpy as np
import pytest

import offmark_oracle as orc
from conftest import GOLDEN, golden_cases

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


C21_TOL = 1e-3
PIXEL_FRAC = 1e-5        # measured 2.6e-7 (profiles/r2_parity_stats.txt)
BITS_FRAC = 1e-4         # measured 0

# every pixel / raw-bit comparison of this module: (kind, test id, differing, compared); written to
# gpurun_out/parity_measured.json at the end of the session so the budgets above stay honest
_MEASURED = []


def _log(kind, bad, n):
    _MEASURED.append((kind, os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], int(bad), int(n)))


@pytest.fixture(scope="module", autouse=True)
def _dump_measured():
    yield
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tot = {}
    for kind, _, bad, n in _MEASURED:
        t = tot.setdefault(kind, [0, 0])
        t[0] += bad
        t[1] += n
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_measured.json"), "w") as f:
            json.dump({"totals": {k: dict(differing=v[0], compared=v[1], frac=v[0] / max(v[1], 1)) for k, v in tot.items()},
                       "worst": sorted(({"kind": k, "test": t, "differing": b, "compared": n} for k, t, b, n in _MEASURED if b),
                                       key=lambda d: -d["differing"] / max(d["compared"], 1))[:40]}, f, indent=1)
    except OSError:
        pass


def oracle_embed_debug(frame, wm, alpha):
    enc = orc.DctEncoderOracle(alpha=alpha)
    enc.read_wm(wm)
    enc.encode(orc.bgr2yuv_f32(frame.astype(np.float32)))
    return enc.debug


def sign_determined_pixels(frame, wm, alpha):
    """Boolean (H, W) mask of pixels whose block has |C21| > tolerance in the oracle."""
    H, W, _ = frame.shape
    ok = np.abs(oracle_embed_debug(frame, wm, alpha)["c21_pre"]) > C21_TOL
    m = np.ones((H, W), bool)
    m[: ok.shape[0] * 8, : ok.shape[1] * 8] = np.kron(ok, np.ones((8, 8), bool))
    return m, int((~ok).sum())


def assert_pixels_close(got, ref, mask=None):
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    if mask is not None:
        d = d[mask]
    if d.size == 0:
        return
    _log("pixels", (d > 0).sum(), d.size)
    assert d.max() <= 1, f"max pixel diff {d.max()}"
    assert (d > 0).sum() <= budget(d.size, PIXEL_FRAC), f"{(d > 0).sum()} of {d.size} samples differ"


def assert_bits_close(got, ref, nblk):
    mism = int((got.reshape(-1) != ref.reshape(-1)).sum())
    _log("raw_bits", mism, nblk)
    assert mism <= budget(nblk, BITS_FRAC), f"{mism} raw bits differ of {nblk}"


def degen(g, counts, n_bits):
    from offmark.degenerator.de_grayscale import DeGrayScale
    from offmark.degenerator.de_shuffler import DeShuffler
    cls = DeGrayScale if bool(g["image_payload"]) else DeShuffler
    return cls(key=int(g["key"])).set_shape(g["payload"].shape).degenerate_counts(counts, n_bits)


@pytest.mark.parametrize("case", golden_cases())
def test_golden_embed_and_detect(eng, case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    frame = g["frame"]
    H, W, _ = frame.shape
    N, nblk = H * W // 64, (H // 8) * (W // 8)
    L = int(np.prod(g["payload"].shape))
    alpha = float(g["alpha"])
    marked = eng.embed(cuda(frame[None]), g["wm"], alpha=alpha)[0].cpu().numpy()
    mask, n_amb = sign_determined_pixels(frame, g["wm"], alpha)
    assert_pixels_close(marked, g["marked"], mask)
    # our own marked frame must decode to the same bits as the reference's on sign-determined blocks
    _, bits_own = eng.detect(cuda(marked[None]), L, alpha=alpha, want_bits=True)
    det = mask[: (H // 8) * 8: 8, : (W // 8) * 8: 8].reshape(-1)
    assert_bits_close(bits_own[0].cpu().numpy()[:nblk][det], g["raw_bits"].reshape(-1)[:nblk][det], nblk)
    # detect the golden vector's marked frame (the reference's logic over the restated cv2 primitives): isolates the detect path
    counts, bits = eng.detect(cuda(g["marked"][None]), L, alpha=alpha, want_bits=True)
    bits = bits[0].cpu().numpy()
    assert bits.shape == (N,)
    assert_bits_close(bits, g["raw_bits"], nblk)
    assert np.array_equal(counts[0].cpu().numpy(), np.array([bits[i::L].sum() for i in range(L)]))
    if nblk >= 4 * L:       # enough redundancy for the vote to be meaningful
        out = degen(g, counts[0].cpu().numpy(), N)
        assert np.array_equal(np.asarray(out).reshape(-1), np.asarray(g["degenerated"]).reshape(-1))


@pytest.mark.parametrize("case", ["syn_240x320_L8_k0_a20", "frame63_crop0_L8_k0_a20", "syn_30x44_L8_k0_a20",
                                  "edge_black_64x64", "edge_white_64x64"])
def test_debug_planes_against_oracle(eng, case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    frame, alpha = g["frame"], float(g["alpha"])
    enc = orc.DctEncoderOracle(alpha=alpha)
    enc.read_wm(g["wm"])
    enc.encode(orc.bgr2yuv_f32(frame.astype(np.float32)))
    d = eng.debug_planes(cuda(frame), alpha=alpha, wm=g["wm"])
    nblk = d["y_dc"].size
    assert np.max(np.abs(d["y_dc"] - enc.debug["ydc"])) <= 1e-3
    assert np.max(np.abs(d["c21_pre"] - enc.debug["c21_pre"])) <= 1e-3
    for k, ref in (("lum", enc.debug["lum"]), ("tex", enc.debug["tex"])):
        bad = np.abs(d[k] - ref) > 1e-9 * np.maximum(1, np.abs(ref))
        # texture ramp carries the float32 rounding of eh: allow 2e-6 relative there
        bad &= np.abs(d[k] - ref) > 2e-6
        _log("mask_" + k, bad.sum(), nblk)
        assert bad.sum() <= budget(nblk, BITS_FRAC, floor=0), (k, int(bad.sum()))
    same = np.abs(d["step"] - alpha * enc.debug["mask"]) <= 1e-4
    amb = np.abs(enc.debug["c21_pre"]) <= C21_TOL
    post_ok = np.where(amb, np.abs(np.abs(d["c21_post"]) - np.abs(enc.debug["c21_post"])) <= 2e-3,
                       np.abs(d["c21_post"] - enc.debug["c21_post"]) <= 2e-3)
    # an exactly-zero coefficient must stay exactly zero (np.sign(0) == 0, dct_encoder.py:33-35: the bit is lost
    # there): wherever the oracle's C21 is an exact zero, so is the kernel's, before and after the quantiser
    zero = enc.debug["c21_pre"] == 0
    assert (d["c21_pre"][zero] == 0).all() and (d["c21_post"][zero] == 0).all()
    assert (enc.debug["c21_post"][zero] == 0).all()
    _log("step_or_c21_post", (~(same & post_ok)).sum(), nblk)
    assert (~(same & post_ok)).sum() <= budget(nblk, BITS_FRAC, floor=0)


@pytest.mark.parametrize("seed", [2000, 2001, 2003])
def test_1080p_frame_against_oracle(eng, seed):
    H, W = 1080, 1920
    frame = orc.synthetic_frame(H, W, seed)
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(wm)
    ref_marked = orc.mark_frame(frame, enc)
    dec = orc.DctDecoderOracle(alpha=20)
    ref_bits = orc.check_frame(ref_marked, dec)
    marked, counts, bits = eng.embed_detect(cuda(frame[None]), wm, L=8, alpha=20, want_bits=True)
    marked = marked[0].cpu().numpy()
    mask, _ = sign_determined_pixels(frame, wm, 20)
    assert_pixels_close(marked, ref_marked, mask)
    # same input to both detectors -> compare on the oracle's marked frame too
    c2, b2 = eng.detect(cuda(ref_marked[None]), 8, alpha=20, want_bits=True)
    assert_bits_close(b2[0].cpu().numpy(), ref_bits, 32400)
    from offmark.degenerator.de_shuffler import DeShuffler
    deg = DeShuffler(key=0).set_shape(P8.shape)
    assert np.array_equal(deg.degenerate_counts(c2[0].cpu().numpy(), 32400), orc.deshuffle(ref_bits, 8, 0))
    assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), 32400), P8)


def test_non_multiple_of_8_and_unaligned_width(eng):
    for (H, W) in [(30, 44), (17, 9), (64, 100), (8, 8)]:
        frame = orc.synthetic_frame(H, W, 77 + H)
        N, nblk = H * W // 64, (H // 8) * (W // 8)
        wm = orc.shuffle_generate(P8, (1, max(N, 1)), 0) if N else np.zeros((1, 1), np.int64)
        if N == 0:
            continue
        enc = orc.DctEncoderOracle(alpha=20)
        enc.read_wm(wm)
        ref = orc.mark_frame(frame, enc)
        got = eng.embed(cuda(frame[None]), wm, alpha=20)[0].cpu().numpy()
        assert_pixels_close(got, ref, sign_determined_pixels(frame, wm, 20)[0])
        assert np.array_equal(got[(H // 8) * 8:], frame[(H // 8) * 8:])          # fringe passes through
        assert np.array_equal(got[:, (W // 8) * 8:], frame[:, (W // 8) * 8:])
        _, bits = eng.detect(cuda(ref[None]), 8, alpha=20, want_bits=True)
        ref_bits = orc.check_frame(ref, orc.DctDecoderOracle(alpha=20))
        assert bits.shape[1] == N == ref_bits.size
        assert_bits_close(bits[0].cpu().numpy(), ref_bits, nblk)
        assert not bits[0, nblk:].any()


def test_batch_properties_full_size(eng):
    """Config-2 shape (300 x 1080p): size-independent properties instead of an oracle run."""
    import torch
    H, W, n = 1080, 1920, 300
    base = torch.from_numpy(np.stack([orc.synthetic_frame(H, W, 2000 + i) for i in range(6)])).cuda()
    frames = base.repeat(n // 6, 1, 1, 1).contiguous()
    # make every frame distinct: rotate rows by a per-frame multiple of 8
    for i in range(n):
        frames[i] = torch.roll(frames[i], shifts=8 * (i // 6), dims=0)
    N = H * W // 64
    payloads = np.array([[int(b) for b in format(s % 256, "08b")] for s in range(1, 9)])     # per-"segment" payloads
    wm = np.stack([orc.shuffle_generate(p, (N,), 0) for p in payloads])
    rows = (np.arange(n) * 8 // n).astype(np.int32)
    out, counts, _ = eng.embed_detect(frames, wm, L=8, alpha=20, wm_row=rows)
    out2 = eng.embed(frames, wm, alpha=20, wm_row=rows)
    assert torch.equal(out, out2)                                  # fused == separate, deterministic
    eng1 = type(eng)(chunk_frames=1)
    assert torch.equal(eng1.embed(frames[:24], wm, alpha=20, wm_row=rows[:24]), out[:24])   # chunking invariance
    inplace = frames[:24].clone()
    eng.embed(inplace, wm, alpha=20, wm_row=rows[:24], out=inplace)
    assert torch.equal(inplace, out[:24])                          # in-place == out-of-place
    c_sep, _ = eng.detect(out, 8, alpha=20)
    assert torch.equal(c_sep, counts)
    from offmark.degenerator.de_shuffler import DeShuffler
    got = DeShuffler(key=0).set_shape((8,)).degenerate_counts(counts.cpu().numpy(), N)
    assert np.array_equal(got, payloads[rows])                     # every frame recovers its own payload
    # marking is small: PSNR > 35 dB, untouched channel 2
    assert torch.equal(out[..., 2], frames[..., 2])
    mse = (out[:12].float() - frames[:12].float()).pow(2).mean().item()
    assert 10 * np.log10(255 ** 2 / mse) > 35


def test_yuv_plugin_boundary(eng):
    from offmark.embed.dct_encoder import DctEncoder
    from offmark.extract.dct_decoder import DctDecoder
    frame = orc.synthetic_frame(240, 320, 1001)
    yuv = orc.bgr2yuv_f32(frame.astype(np.float32))
    wm = orc.shuffle_generate(P8, (1, 1200), 0)
    enc = DctEncoder(alpha=20)
    assert enc.wm_capacity(yuv.shape) == (1, 1200)
    enc.read_wm(wm)
    ref_enc = orc.DctEncoderOracle(alpha=20)
    ref_enc.read_wm(wm)
    ref = ref_enc.encode(yuv.copy())
    arg = yuv.copy()
    got = enc.encode(arg)
    assert got is arg                                              # mutates and returns its input
    assert np.array_equal(got[:, :, 0], yuv[:, :, 0]) and np.array_equal(got[:, :, 2], yuv[:, :, 2])
    close = np.abs(got[:, :, 1] - ref[:, :, 1]) <= 2e-3
    assert (~close).sum() <= 64 * budget(1200, BITS_FRAC)          # a flipped block moves its 64 samples
    bits = DctDecoder(alpha=20).decode(ref)
    ref_bits = orc.DctDecoderOracle(alpha=20).decode(ref)
    assert bits.dtype == np.float64 and bits.shape == ref_bits.shape == (1, 1200)
    assert_bits_close(bits, ref_bits, 1200)
    y = yuv[:, :, 0]
    lum_bad = np.abs(enc.luminance_mask(y) - orc.luminance_mask_vec(y)) > 1e-6
    tex_bad = np.abs(enc.texture_mask(y) - orc.texture_mask_vec(y)) > 2e-6
    _log("mask_lum", lum_bad.sum(), 1200)
    _log("mask_tex", tex_bad.sum(), 1200)
    assert lum_bad.sum() == 0 and tex_bad.sum() == 0


def test_abi_error_codes(eng):
    import torch
    from offmark import _hip
    lib = _hip.load()
    f = torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device="cuda")
    ws = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
    wm = torch.zeros(4, dtype=torch.uint8, device="cuda")
    s = _hip.current_stream()
    assert lib.ofmk_embed_rgb8(None, f.data_ptr(), 1, 16, 16, wm.data_ptr(), 1, None, 20.0, 0, ws.data_ptr(), ws.numel(), s, None) == -1
    assert lib.ofmk_embed_rgb8(f.data_ptr(), f.data_ptr(), 1, 4, 16, wm.data_ptr(), 1, None, 20.0, 0, ws.data_ptr(), ws.numel(), s, None) == -1
    assert b"at least 8" in lib.ofmk_last_error()
    assert lib.ofmk_embed_rgb8(f.data_ptr(), f.data_ptr(), 1, 16, 16, wm.data_ptr(), 1, None, 20.0, 0, ws.data_ptr(), 8, s, None) == -2
    assert lib.ofmk_detect_rgb8(f.data_ptr(), 1, 16, 16, 0, 20.0, ws.data_ptr(), None, 0, ws.data_ptr(), ws.numel(), s, None) == -1
    assert lib.ofmk_workspace_bytes(0, 16, 16) == 0 and lib.ofmk_workspace_bytes(1, 16, 16) > 0
    with pytest.raises(_hip.HipError):
        _hip.check(-1)


def test_fused_verify_kernel_equals_separate_kernels(eng):
    """ofmk_embed_detect_rgb8: the fused mark+analyze kernel and the two separate kernels must agree
    bit for bit (same arithmetic on the same rounded pixels)."""
    import torch
    from offmark import _hip
    frames = cuda(np.stack([orc.synthetic_frame(240, 320, 1001 + i) for i in range(5)] ))
    nat = np.load(os.path.join(GOLDEN, "frame63_crop_qr_k0_a20.npz"))["frame"]
    wm = orc.shuffle_generate(P8, (1, 1200), 0)
    o1, c1, b1 = eng.embed_detect(frames, wm, L=8, want_bits=True)
    sep = type(eng)(opts=_hip.Opts(_hip.F_SEPARATE_DETECT, 0, None))       # per-engine option, no process-wide switch
    o2, c2, b2 = sep.embed_detect(frames, wm, L=8, want_bits=True)
    assert torch.equal(o1, o2) and torch.equal(c1, c2) and torch.equal(b1, b2)
    wmn = orc.shuffle_generate(P8, (1, nat.shape[0] * nat.shape[1] // 64), 0)
    o3, c3, b3 = eng.embed_detect(cuda(nat[None]), wmn, L=8, want_bits=True)
    c4, b4 = eng.detect(o3, 8, want_bits=True)
    assert torch.equal(c3, c4) and torch.equal(b3, b4)


def test_device_payload_epilogue_equals_host(eng):
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    rng = np.random.default_rng(9)
    for (L, N, key) in [(8, 32400, 0), (5, 37, 7), (441, 1536, 0), (13, 6, 3), (300, 32400, 1)]:
        deg = DeShuffler(key=key).set_shape((L,))
        lens = np.array([max(0, (N - i + L - 1) // L) if i < N else 0 for i in range(L)])
        counts = (rng.random((17, L)) * (lens + 0.999)).astype(np.int32)
        counts = np.minimum(counts, lens).astype(np.int32)
        with np.errstate(all="ignore"):
            ref = deg.degenerate_counts(counts, N)
        got = eng.payloads(torch.from_numpy(counts).cuda(), N, deg.payload_idx).cpu().numpy()
        assert np.array_equal(got, ref), (L, N)


def test_config1_pipeline_on_gpu_209_frames(eng):
    """mark.py / detect.py logic on an in-memory stand-in for the bundled 320x240, 209-frame clip."""
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.embed.dct_encoder import DctEncoder
    from offmark.extract.dct_decoder import DctDecoder
    from offmark.generator.shuffler import Shuffler
    from offmark.video.embedder import Embedder
    from offmark.video.extractor import Extractor
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark.video.frame_writer import ArrayFrameWriter
    base = [orc.synthetic_frame(240, 320, 1001 + i) for i in range(11)]
    frames = np.stack([np.roll(base[i % 11], 8 * (i // 11), axis=1) for i in range(209)])
    r, w = ArrayFrameReader(frames), ArrayFrameWriter()
    frame_embedder = DctEncoder()
    capacity = frame_embedder.wm_capacity((r.height, r.width, 3))
    frame_embedder.read_wm(Shuffler(key=0).generate_wm(P8, capacity))
    emb = Embedder(r, frame_embedder, w, batch_frames=50)
    emb.start()
    assert emb.frames_marked == 209 and len(w.frames) == 209 and r.closed and w.closed
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(Shuffler(key=0).generate_wm(P8, capacity))
    for i in (0, 57, 208):
        mask, _ = sign_determined_pixels(frames[i], frame_embedder.wm[None], 20)
        assert_pixels_close(w.frames[i], orc.mark_frame(frames[i], enc), mask)
    ex = Extractor(ArrayFrameReader(w.frames), DctDecoder(), DeShuffler(key=0).set_shape(P8.shape), batch_frames=64)
    ex.start()
    assert len(ex.patterns) == 209 and all(np.array_equal(p, P8) for p in ex.patterns)
    assert np.array_equal(ex.most_common()[0], P8)


def test_4k_frame_against_oracle(eng):
    H, W = 2160, 3840
    frame = orc.synthetic_frame(H, W, 3001)
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(wm)
    ref_marked = orc.mark_frame(frame, enc)
    marked, counts, bits = eng.embed_detect(cuda(frame[None]), wm, L=8, want_bits=True)
    mask = np.ones((H, W), bool)
    ok = np.abs(enc.debug["c21_pre"]) > C21_TOL
    mask[:] = np.kron(ok, np.ones((8, 8), bool))
    assert_pixels_close(marked[0].cpu().numpy(), ref_marked, mask)
    ref_bits = orc.check_frame(ref_marked, orc.DctDecoderOracle(alpha=20))
    _, b2 = eng.detect(cuda(ref_marked[None]), 8, want_bits=True)
    assert_bits_close(b2[0].cpu().numpy(), ref_bits, 129600)
    from offmark.degenerator.de_shuffler import DeShuffler
    assert np.array_equal(DeShuffler(key=0).set_shape((8,)).degenerate_counts(counts[0].cpu().numpy(), 129600), P8)


def test_4k_multi_chunk_batch(eng):
    """BASELINE config 3 shape (4K frames processed in several internal chunks): 100 frames with chunk_frames=40
    (balanced: chunks of 34, 34, 32) must equal the single-chunk result bit for bit, every frame must recover its own
    payload, and chunk boundaries must not leak state (per-frame means, counts)."""
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.synthetic import synthetic_frames
    H, W, n = 2160, 3840, 100
    N = H * W // 64
    base = synthetic_frames(10, H, W, seed=3100)
    frames = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    for i in range(n):                                              # distinct frames: rolled by whole blocks
        frames[i] = torch.roll(base[i % 10], shifts=(8 * (i // 10), 16 * (i // 10)), dims=(0, 1))
    payloads = np.array([[int(b) for b in format(s, "08b")] for s in (0x65, 0x9a, 0x3c)])
    wm = np.stack([orc.shuffle_generate(p, (N,), 0) for p in payloads])
    rows = (np.arange(n) % 3).astype(np.int32)
    chunked = type(eng)(chunk_frames=40)
    out_c, counts_c, _ = chunked.embed_detect(frames, wm, L=8, wm_row=rows)
    whole = type(eng)(chunk_frames=n)
    out_w, counts_w, _ = whole.embed_detect(frames, wm, L=8, wm_row=rows)
    assert torch.equal(out_c, out_w) and torch.equal(counts_c, counts_w)
    got = DeShuffler(key=0).set_shape((8,)).degenerate_counts(counts_c.cpu().numpy(), N)
    assert np.array_equal(got, payloads[rows])
    c_det, _ = chunked.detect(out_c, 8)                              # stand-alone detect, chunked, agrees with the fused verify
    assert torch.equal(c_det, counts_c)
    assert torch.equal(out_c[..., 2], frames[..., 2])
    # one frame of the batch against the oracle (a frame in the LAST, shorter chunk)
    k = 93
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(wm[rows[k]][None])
    f_host = frames[k].cpu().numpy()
    ref = orc.mark_frame(f_host, enc)
    okb = np.abs(enc.debug["c21_pre"]) > C21_TOL
    assert_pixels_close(out_c[k].cpu().numpy(), ref, np.kron(okb, np.ones((8, 8), bool)))


def test_tile_orders_are_bit_identical_and_the_xcc_probe_reads_the_deal(eng):
    """The fused mark kernel's tile order (ofmk_opts: OFMK_F_LINEAR_TILES, xcds) is a pure permutation of the workgroups:
    linear, XCD-aware over 8 and over other XCD counts (padding workgroups, a count that does not divide the grid) must give
    the same marked frames, counts and bits.  ofmk_probe_xcc must report a sane deal; the default is the library's static rule
    on the launch size and an engine can force either order."""
    import torch
    from offmark import _hip, engine as E
    from offmark.synthetic import synthetic_frames
    H, W, n = 360, 648, 37                                   # 45 x 81 blocks: 15 tiles per frame (ragged last tile), 555 tiles
    frames = synthetic_frames(n, H, W, seed=77)
    N = H * W // 64
    wm = np.stack([orc.shuffle_generate(P8, (N,), 0), orc.shuffle_generate(1 - P8, (N,), 0)])
    rows = (np.arange(n) % 2).astype(np.int32)
    ref = None
    for flags, xcds in ((0, 0), (_hip.F_LINEAR_TILES, 0), (0, 1), (0, 4), (0, 7), (0, 8), (0, 64), (_hip.F_SEPARATE_DETECT, 5)):
        e = type(eng)(opts=_hip.Opts(flags | (0 if flags & _hip.F_LINEAR_TILES else _hip.F_XCD_TILES), xcds, None))
        got = e.embed_detect(frames, wm, L=8, wm_row=rows, want_bits=True)
        plain = e.embed(frames, wm, wm_row=rows)
        assert torch.equal(plain, got[0])
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), (flags, xcds)
    lib = _hip.load()
    ws = eng.workspace(H, W, n)
    assert lib.ofmk_embed_rgb8(frames.data_ptr(), ref[0].data_ptr(), n, H, W, cuda(wm.astype(np.uint8)).data_ptr(), 2, None, 20.0, 0,
                               ws.data_ptr(), ws.numel(), _hip.current_stream(), _hip.Opts(0, 65, None)) == -1
    assert b"xcds" in lib.ofmk_last_error()
    deal = E.probe_xcc_deal()
    assert 1 <= deal["xcds"] <= 16 and len(deal["ids_by_residue"]) == deal["xcds"] and 0.0 < deal["round_robin_fraction"] <= 1.0
    print("xcc deal:", deal)
    # default policy: no measurement -- the library's static rule on the launch size (small launches linear, large ones XCD-aware)
    big = synthetic_frames(48, 1080, 1920, seed=78)
    wm_big = orc.shuffle_generate(P8, (32400,), 0)[None]
    auto = type(eng)()
    a = auto.embed_detect(big, wm_big, L=8)
    assert auto.tile_order == "linear" and auto.tile_order_info["policy"] == "static rule"       # 48 frames of 1080p < 192
    assert E.static_tile_order(192 * 1080 * 1920 * 3) == "xcd" and E.static_tile_order(191 * 1080 * 1920 * 3) == "linear"
    for forced in ("xcd", "linear"):
        b = type(eng)(tile_order=forced).embed_detect(big, wm_big, L=8)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    both = _hip.Opts(_hip.F_LINEAR_TILES | _hip.F_XCD_TILES, 0, None)
    assert lib.ofmk_embed_rgb8(frames.data_ptr(), ref[0].data_ptr(), n, H, W, cuda(wm.astype(np.uint8)).data_ptr(), 2, None, 20.0, 0,
                               ws.data_ptr(), ws.numel(), _hip.current_stream(), both) == -1
    with pytest.raises(ValueError, match="tile_order"):
        type(eng)(tile_order="calibrate")                             # rounds 4-5's opt-in measuring mode is gone
    # round 6: in XCD order the NON-fused mark kernel is the HOLD form (the 8 marked rows stored at the end, dct_kernels.hiph:
    # mark_rows) -- also for widths that are no multiple of 8 (byte-wise loads and stores), ragged last tiles, one-tile frames and
    # in place: same bytes as the row-by-row form in linear order
    for (h, w, m) in [(64, 100, 9), (30, 44, 5), (17, 9, 3), (8, 8, 70), (360, 652, 11)]:
        fr = synthetic_frames(m, h, w, seed=500 + w)
        wmq = np.stack([orc.shuffle_generate(P8, (h * w // 64,), 0), orc.shuffle_generate(1 - P8, (h * w // 64,), 0)])
        rq = (np.arange(m) % 2).astype(np.int32)
        lin = type(eng)(tile_order="linear").embed(fr, wmq, wm_row=rq)
        hold = type(eng)(tile_order="xcd").embed(fr, wmq, wm_row=rq)
        assert torch.equal(lin, hold), (h, w, m)
        buf = fr.clone()
        type(eng)(tile_order="xcd").embed(buf, wmq, wm_row=rq, out=buf)
        assert torch.equal(buf, lin), (h, w, m)


def test_two_threads_two_engines(eng):
    """include/offmark_hip.h: the library has no mutable state besides the per-thread error text.  Two host threads
    drive two engines (own workspace, stream, options and timing object) at the same time; each must get exactly
    what it gets alone, the per-thread error texts must not mix, and each timing object must see only its own
    engine's launches."""
    import threading
    import torch
    from offmark import _hip
    from offmark.synthetic import synthetic_frames
    lib = _hip.load()
    H, W, n = 240, 320, 24
    N = H * W // 64
    fa, fb = synthetic_frames(n, H, W, seed=11), synthetic_frames(n, 120, 200, seed=12)
    wma = orc.shuffle_generate(P8, (1, N), 0)
    wmb = orc.shuffle_generate(P8[::-1].copy(), (1, 120 * 200 // 64), 5)
    ref_a = eng.embed_detect(fa, wma, L=8, want_bits=True)
    ref_b = type(eng)(opts=_hip.Opts(_hip.F_SEPARATE_DETECT, 0, None)).embed_detect(fb, wmb, L=8, alpha=10, want_bits=True)
    torch.cuda.synchronize()
    results, errors = {}, {}

    def work(name, frames, wm, alpha, flags, bad_h, bad_ws, want_rc, want_txt):
        try:
            torch.cuda.set_device(0)
            timing = _hip.Timing(64 * 8)
            e = type(eng)(opts=timing.opts(flags))
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(20):
                    got = e.embed_detect(frames, wm, L=8, alpha=alpha, want_bits=True)
                    # a failing call on this thread: its error text must be this thread's own
                    rc = lib.ofmk_embed_rgb8(frames.data_ptr(), frames.data_ptr(), 1, bad_h, 16, frames.data_ptr(), 1, None,
                                             20.0, 0, frames.data_ptr(), bad_ws, stream.cuda_stream, None)
                    txt = lib.ofmk_last_error()
                    assert rc == want_rc and want_txt in txt, (rc, txt)
            stream.synchronize()
            results[name] = (got, timing.collect())
            timing.close()
        except Exception as exc:                                     # surfaced by the main thread
            errors[name] = exc

    # both failing calls are rejected before anything is enqueued (H < 8; a workspace of 8 bytes)
    ta = threading.Thread(target=work, args=("a", fa, wma, 20, 0, 4, 1 << 20, -1, b"at least 8"))
    tb = threading.Thread(target=work, args=("b", fb, wmb, 10, _hip.F_SEPARATE_DETECT, 16, 8, -2, b"workspace smaller"))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errors, errors
    for name, ref in (("a", ref_a), ("b", ref_b)):
        got, _ = results[name]
        assert all(torch.equal(x, y) for x, y in zip(got, ref)), name
    ka, kb = results["a"][1], results["b"][1]
    assert ka["mark_fused"]["launches"] == 20 and ka["mark"]["launches"] == 0 and ka["analyze"]["launches"] == 20
    assert kb["mark"]["launches"] == 20 and kb["mark_fused"]["launches"] == 0 and kb["analyze"]["launches"] == 40


def test_stage_entry_points_and_copy(eng):
    import torch
    from offmark import _hip
    lib = _hip.load()
    frames = cuda(np.stack([orc.synthetic_frame(64, 96, 5 + i) for i in range(3)]))
    wm = orc.shuffle_generate(P8, (1, 96), 0)
    ref = eng.embed(frames, wm)
    ws = eng.workspace(64, 96, 3)
    out = torch.empty_like(frames)
    s = _hip.current_stream()
    wm_dev = cuda(wm.astype(np.uint8))
    _hip.check(lib.ofmk_stage_analyze_rgb8(frames.data_ptr(), 3, 64, 96, ws.data_ptr(), ws.numel(), s, None))
    _hip.check(lib.ofmk_stage_mark_rgb8(frames.data_ptr(), out.data_ptr(), 3, 64, 96, wm_dev.data_ptr(), 20.0, 0,
                                        ws.data_ptr(), ws.numel(), s, None))
    assert torch.equal(out, ref)
    a = torch.arange(1 << 20, dtype=torch.int32, device="cuda")
    b = torch.zeros_like(a)
    _hip.check(lib.ofmk_hbm_copy(a.data_ptr(), b.data_ptr(), a.numel() * 4, s))
    assert torch.equal(a, b)
    assert lib.ofmk_hbm_copy(a.data_ptr(), b.data_ptr(), 7, s) == -1
    for nbytes in (16, 16 * 1023, 16 * 1025, 16 * 5000 + 16):            # ragged last span of the copy
        b.zero_()
        _hip.check(lib.ofmk_hbm_copy(a.data_ptr(), b.data_ptr(), nbytes, s))
        assert torch.equal(a[: nbytes // 4], b[: nbytes // 4]) and not b[nbytes // 4:].any()
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    _hip.check(lib.ofmk_hbm_read(a.data_ptr(), a.numel() * 4, sink.data_ptr(), s))
    _hip.check(lib.ofmk_hbm_read(a.data_ptr(), 16 * 2049, sink.data_ptr(), s))
    torch.cuda.synchronize()
    assert lib.ofmk_hbm_read(a.data_ptr(), 8, sink.data_ptr(), s) == -1


def test_config4_segments_with_own_payloads(eng):
    """BASELINE config 4 shape on one GPU: 8 segments x 48 frames of 1080p, segment s carries
    format(s % 256, '08b') (segment_mark_detect_hls.py:42-55); per-segment Counter vote must return it."""
    import torch
    from offmark import fingerprint as fp
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.vote import vote_segments
    from offmark.synthetic import synthetic_frames
    H, W, S, F = 1080, 1920, 8, 48
    N = H * W // 64
    frames = synthetic_frames(S * F, H, W, seed=4000)
    payloads = np.stack([fp.payload_for_segment(s + 1) for s in range(S)])
    wm = np.stack([orc.shuffle_generate(p, (N,), 0) for p in payloads])
    seg = np.repeat(np.arange(S), F)
    _, counts, _ = eng.embed_detect(frames, wm, L=8, wm_row=seg.astype(np.int32))
    deg = DeShuffler(key=0).set_shape((8,))
    per_frame = eng.payloads(counts, N, deg.payload_idx).cpu().numpy()
    assert np.array_equal(per_frame, deg.degenerate_counts(counts.cpu().numpy(), N))
    votes = vote_segments(per_frame, seg)
    for s in range(S):
        assert np.array_equal(votes[s][0], payloads[s]) and votes[s][1] == 1.0


def test_config5_leak_identification_with_build_defined_attacks(eng):
    """BASELINE config 5 shape: 8 segments x 3 copies, payload = segment(4b)||copy(4b)
    (mark_video_to_hls.py:38-43); a leak picks one copy per segment (generate_leak.py:59-108) and the
    detector must return the copy sequence (detect_watermarks.py:345-364).  The reference has no
    attacks; the ones here are build-defined tensor ops.  Only 'none' and 'requantisation noise' are
    gated; scaling and cropping break the 8x8 grid and are merely reported."""
    import torch
    from offmark import fingerprint as fp
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.vote import vote_segments
    from offmark.synthetic import synthetic_frames
    H, W, S, C, F = 240, 320, 8, 3, 12
    N = H * W // 64
    src = synthetic_frames(S * F, H, W, seed=5000)
    deg = DeShuffler(key=0).set_shape((8,))
    wm_table = np.stack([orc.shuffle_generate(fp.payload_for_segment(s, c), (N,), 0) for s in range(S) for c in range(C)])
    seg = np.repeat(np.arange(S), F)
    copies = [eng.embed(src, wm_table, wm_row=(seg * C + c).astype(np.int32)) for c in range(C)]
    leak_pattern = "01201201"
    chosen = fp.select_copies(leak_pattern, S, C)
    leak = torch.cat([copies[chosen[s]][s * F:(s + 1) * F] for s in range(S)])

    def identify(frames):
        counts, _ = eng.detect(frames.contiguous(), 8)
        votes = vote_segments(eng.payloads(counts, N, deg.payload_idx).cpu().numpy(), seg)
        return fp.identify_copies(votes)

    g = torch.Generator(device="cuda").manual_seed(1)
    noisy = (leak.float() + 2.0 * torch.randn(leak.shape, device="cuda", generator=g)).round().clamp(0, 255).to(torch.uint8)
    assert identify(leak) == chosen
    assert identify(noisy) == chosen
    x = leak.permute(0, 3, 1, 2).float()
    small = torch.nn.functional.interpolate(x, size=(H * 2 // 3, W * 2 // 3), mode="bilinear", align_corners=False)
    scaled = torch.nn.functional.interpolate(small, size=(H, W), mode="bilinear", align_corners=False)
    cropped = torch.nn.functional.interpolate(x[:, :, 16:-16, 16:-16], size=(H, W), mode="bilinear", align_corners=False)
    for name, t in (("scale 2/3", scaled), ("crop 16", cropped)):
        got = identify(t.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1))
        print(f"attack {name}: recovered {sum(a == b for a, b in zip(got, chosen))}/{S} copies (not gated)")
    # 4:2:0 chroma subsampling round trip (what FileEncoder's yuv420p output does to every marked frame,
    # frame_writer.py:34): average U,V over 2x2, replicate back.  The reference's own robustness bar is
    # ">= 75 % of segments keep their payload" through an HLS re-encode (segment_mark_detect_hls.py:500).
    k = torch.tensor([[0.114, 0.587, 0.299], [-0.114 * 0.492 + 0.492, -0.587 * 0.492, -0.299 * 0.492],
                      [-0.114 * 0.877, -0.587 * 0.877, -0.299 * 0.877 + 0.877]], device="cuda")
    yuv = torch.einsum("kc,nchw->nkhw", k, x)
    uv = torch.nn.functional.avg_pool2d(yuv[:, 1:], 2).repeat_interleave(2, 2).repeat_interleave(2, 3)
    y_, u_, v_ = yuv[:, 0], uv[:, 0], uv[:, 1]
    back = torch.stack([y_ + 2.032 * u_, y_ - 0.395 * u_ - 0.581 * v_, y_ + 1.140 * v_], dim=1)
    got = identify(back.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1))
    kept = sum(a == b for a, b in zip(got, chosen))
    print(f"attack 4:2:0 chroma round trip: recovered {kept}/{S} copies")
    assert kept >= 0.75 * S


def test_c_abi_calls_are_graph_capturable(eng):
    """include/offmark_hip.h promises no allocation and no synchronisation inside the compute calls:
    a whole embed+detect+payload step must capture into a HIP graph and replay with identical results."""
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.synthetic import synthetic_frames
    H, W, n = 240, 320, 16
    N = H * W // 64
    frames = synthetic_frames(n, H, W, seed=77)
    wm = cuda(orc.shuffle_generate(P8, (1, N), 0).astype(np.uint8))
    perm = torch.as_tensor(DeShuffler(key=0).set_shape((8,)).payload_idx, dtype=torch.int32).cuda()
    out = torch.empty_like(frames)
    ref_out, ref_counts, _ = eng.embed_detect(frames, wm, L=8)
    ref_pay = eng.payloads(ref_counts, N, perm)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        eng.embed_detect(frames, wm, L=8, out=out)          # warm-up on the capture stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            _, counts, _ = eng.embed_detect(frames, wm, L=8, out=out)
            pay = eng.payloads(counts, N, perm)
    out.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out) and torch.equal(counts, ref_counts) and torch.equal(pay, ref_pay)
    del graph, counts, pay
    import gc
    gc.collect()
    torch.cuda.synchronize()


@pytest.mark.parametrize("codec", ["dct", "dct-detect", "svd4", "svd8", "planar"])
def test_several_steps_in_one_graph_replay_like_eager(eng, codec):
    """Three steps with three DIFFERENT batches captured into ONE hipGraph (bench.py's grouped steps: a 48-frame segment is too
    little work to issue step by step), replayed four times: every replay of every step must equal the eager result bit for bit
    -- marked frames, counts, payloads.  The steps share the engine's workspace (mean accumulators) and, through torch's
    caching allocator, their counts buffer: exactly the shape that broke while the library zeroed buffers with hipMemsetAsync
    (memset nodes of one buffer replay out of order on ROCm 7.2 from the second replay on: profiles/r4_graph_memset_order.txt;
    the library now zero-fills with a kernel)."""
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.synthetic import synthetic_frames
    H, W, n, G = 240, 320, 12, 3
    N = H * W // 64
    blk = 8 if codec == "svd8" else 4
    nb = type(eng).svd_bits_per_frame(H, W, blk) if codec.startswith("svd") else N
    wm = cuda(np.stack([orc.shuffle_generate(P8, (N,), 0), orc.shuffle_generate(1 - P8, (N,), 0)]).astype(np.uint8))
    rows = [cuda(((np.arange(n) + g) % 2).astype(np.int32)) for g in range(G)]
    perm = torch.as_tensor(DeShuffler(key=0).set_shape((8,)).payload_idx, dtype=torch.int32).cuda()
    batches = [synthetic_frames(n, H, W, seed=900 + g) for g in range(G)]
    if codec == "planar":
        batches = [eng.rgb_to_yuv420(b) for b in batches]
    if codec == "dct-detect":
        batches = [eng.embed(b, wm, wm_row=rows[g]) for g, b in enumerate(batches)]
    outs = [torch.zeros_like(b) for b in batches]
    pays = torch.zeros((G, n, 8), dtype=torch.uint8, device="cuda")
    keep = [None] * G

    def one(g, keep_counts):
        if codec == "dct":
            _, c, _ = eng.embed_detect(batches[g], wm, L=8, wm_row=rows[g], out=outs[g])
        elif codec == "dct-detect":
            c, _ = eng.detect(batches[g], 8)
        elif codec == "planar":
            _, c, _ = eng.embed_detect_yuv420(batches[g], H, W, wm, 8, wm_row=rows[g], out=outs[g])
        else:
            _, c, _ = eng.svd_embed_detect(batches[g], wm, L=8, wm_row=rows[g], out=outs[g], blk=blk)
        eng.payloads(c, nb, perm, out=pays[g])
        if keep_counts:
            keep[g] = c.clone()

    for g in range(G):
        one(g, True)
    torch.cuda.synchronize()
    ref = ([o.clone() for o in outs], pays.clone(), [k.clone() for k in keep])
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for g in range(G):
            one(g, False)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for g in range(G):
                one(g, False)                      # counts freed on return: the next step's allocation reuses the block
    torch.cuda.synchronize()
    for rep in range(4):
        pays.zero_()
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(pays, ref[1]), (codec, rep, [bool(torch.equal(pays[g], ref[1][g])) for g in range(G)])
        if codec != "dct-detect":
            assert all(torch.equal(a, b) for a, b in zip(outs, ref[0])), (codec, rep)
    assert (pays[0].cpu().numpy() == np.where((np.arange(n) % 2)[:, None] == 0, P8, 1 - P8)).all()
    del graph                                      # release the graph and its private pool here, not at some later collection
    import gc
    gc.collect()
    torch.cuda.synchronize()


@pytest.mark.parametrize("codec", ["dct", "dwtdctsvd"])
def test_mark_copies_sidecars_and_leak_identification(eng, codec, tmp_path):
    """mark_video_to_hls.py's flow (N copies per segment, verify, JSON sidecars) followed by
    detect_watermarks.py's mapping branch on a leak assembled from the copies."""
    import json
    import torch
    from offmark import fingerprint as fp
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.vote import vote_segments
    from offmark.synthetic import synthetic_frames
    if codec == "dct":
        from offmark.embed.dct_encoder import DctEncoder as Enc
        from offmark.extract.dct_decoder import DctDecoder as Dec
    else:
        from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder as Enc
        from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder as Dec
    H, W, S, C, F = 240, 320, 6, 3, 10
    frames = synthetic_frames(S * F, H, W, seed=6000)
    # segments 1..S: the reference's payload for (segment 0, copy 0) is all zeros, which its own mid-range
    # threshold cannot decode reliably (any raw bit error outvotes a constant payload, de_shuffler.py:20-21)
    seg = np.repeat(np.arange(1, S + 1), F)
    enc, dec = Enc(), Dec()
    copies, side = fp.mark_segment_copies(enc, dec, frames, seg, C)
    assert len(copies) == C and not side["failed_segments"]
    assert side["segment_copies"]["total_marked_segments"] == S * C
    assert side["segment_payloads"]["4_2"] == fp.payload_for_segment(4, 2).tolist()
    paths = fp.write_sidecars(str(tmp_path), side)
    assert [os.path.basename(p) for p in paths] == ["segment_payloads.json", "segment_copies.json"]
    payloads = json.load(open(paths[0]))
    chosen = fp.select_copies("120210", S, C)
    leak = torch.cat([copies[chosen[s]][s * F:(s + 1) * F] for s in range(S)])
    counts, _ = dec.decode_frames_u8(leak.contiguous(), 8)
    votes = vote_segments(DeShuffler(key=0).set_shape((8,)).degenerate_counts(counts.cpu().numpy(), H * W // 64), seg)
    rows = fp.identify_copies_with_payloads(votes, payloads, C)
    assert [r["detected_copy_index"] for r in rows] == chosen and all(r["success"] for r in rows)
    assert fp.identify_copies(votes) == chosen


def test_long_and_oversized_payloads_and_empty_inputs(eng):
    """Payload longer than the LDS histogram (global-atomic path), payload longer than the frame's capacity
    (trailing positions have empty slices -> numpy's nan semantics), and empty/invalid inputs."""
    import torch
    from offmark import _hip
    from offmark.degenerator.de_shuffler import DeShuffler
    lib = _hip.load()
    H, W = 240, 320
    N = H * W // 64
    frame = orc.synthetic_frame(H, W, 1001)
    for L in (3000, 1200, 2048, 2049):
        rng = np.random.default_rng(L)
        payload = rng.integers(0, 2, size=L)
        wm = orc.shuffle_generate(payload, (1, N), 3)
        marked, counts, bits = eng.embed_detect(cuda(frame[None]), wm, L=L, want_bits=True)
        b = bits[0].cpu().numpy()
        assert np.array_equal(counts[0].cpu().numpy(), np.array([b[i::L].sum() for i in range(L)]))
        deg = DeShuffler(key=3).set_shape((L,))
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = orc.deshuffle(b.astype(np.float64), L, 3)
        assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), N), ref)
        perm = torch.as_tensor(deg.payload_idx, dtype=torch.int32).cuda()
        assert np.array_equal(eng.payloads(counts, N, perm)[0].cpu().numpy(), ref)
        if L <= N:
            assert (ref == payload).mean() > 0.95          # one repeat per bit at most: nearly all survive
    f = cuda(frame[None])
    ws = eng.workspace(H, W, 1)
    wm = cuda(orc.shuffle_generate(P8, (1, N), 0).astype(np.uint8))
    s = _hip.current_stream()
    assert lib.ofmk_embed_rgb8(f.data_ptr(), f.data_ptr(), 0, H, W, wm.data_ptr(), 1, None, 20.0, 0, ws.data_ptr(), ws.numel(), s, None) == -1
    assert lib.ofmk_svd_embed_rgb8(f.data_ptr(), f.data_ptr(), 1, H, W, wm.data_ptr(), 1, None, _hip.scales3(0), 4, s, None) == -1
    assert lib.ofmk_svd_detect_rgb8(f.data_ptr(), 1, H, W, 8, _hip.scales3(15), 4, None, None, s, None) == -1
    assert lib.ofmk_svd_embed_rgb8(f.data_ptr(), f.data_ptr(), 1, H, W, wm.data_ptr(), 1, None, _hip.scales3(15), 5, s, None) == -1       # blk: 4 or 8
    assert lib.ofmk_payloads_from_counts(None, 1, 8, N, None, None, s, None) == -1
    unaligned = torch.empty(ws.numel() + 1, dtype=torch.uint8, device="cuda")[1:]
    assert lib.ofmk_embed_rgb8(f.data_ptr(), f.data_ptr(), 1, H, W, wm.data_ptr(), 1, None, 20.0, 0, unaligned.data_ptr(), ws.numel(), s, None) == -1
    assert b"256-byte aligned" in lib.ofmk_last_error()


def test_wm_row_map_is_range_checked_on_the_host(eng):
    frames = cuda(np.stack([orc.synthetic_frame(64, 96, 1)] * 2))
    wm = np.stack([orc.shuffle_generate(P8, (96,), 0)] * 2)
    eng.embed(frames, wm, wm_row=np.array([1, 0]))
    for bad in (np.array([0, 2]), np.array([-1, 0]), [0, 5]):
        with pytest.raises(ValueError, match="wm_row"):
            eng.embed(frames, wm, wm_row=bad)
    with pytest.raises(ValueError, match="one entry per frame"):
        eng.embed(frames, wm, wm_row=np.array([0]))


def test_inplace_and_unaligned_frame_pointers(eng):
    """In-place marking for both codecs, and frame pointers that are not 8-byte aligned (generic byte path)."""
    import torch
    frames = np.stack([orc.synthetic_frame(64, 96, 50 + i) for i in range(3)])
    wm = orc.shuffle_generate(P8, (1, 96), 0)
    ref = eng.embed(cuda(frames), wm)
    buf = torch.empty(frames.size + 3, dtype=torch.uint8, device="cuda")
    view = buf[3:].view(3, 64, 96, 3)                       # data_ptr % 8 == 3
    view.copy_(cuda(frames))
    assert view.data_ptr() % 8 != 0
    out = torch.empty(frames.size + 5, dtype=torch.uint8, device="cuda")[5:].view(3, 64, 96, 3)
    eng.embed(view, wm, out=out)
    assert torch.equal(out, ref)
    eng.embed(view, wm, out=view)                           # in place, unaligned
    assert torch.equal(view, ref)
    sref = eng.svd_embed(cuda(frames), wm)
    inpl = cuda(frames)
    eng.svd_embed(inpl, wm, out=inpl)
    assert torch.equal(inpl, sref)
    c1, b1 = eng.detect(ref, 8, want_bits=True)
    c2, b2 = eng.detect(view, 8, want_bits=True)
    assert torch.equal(c1, c2) and torch.equal(b1, b2)


def test_full_natural_1080p_frame(eng):
    """The reference's own 1920x1080 JPEG frame (chroma-subsampled content: ~15 % of its blocks are chroma-flat,
    i.e. sign-ambiguous) through the HIP path, against the oracle."""
    from conftest import natural_frame
    from offmark.degenerator.de_shuffler import DeShuffler
    nat = natural_frame()
    wm = orc.shuffle_generate(P8, (1, 32400), 0)
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(wm)
    ref = orc.mark_frame(nat, enc)
    ref_bits = orc.check_frame(ref, orc.DctDecoderOracle(alpha=20)).reshape(-1)
    marked, counts, bits = eng.embed_detect(cuda(nat[None]), wm, L=8, want_bits=True)
    mask, n_amb = sign_determined_pixels(nat, wm, 20)
    assert 0.02 < n_amb / 32400 < 0.5
    assert_pixels_close(marked[0].cpu().numpy(), ref, mask)
    c2, b2 = eng.detect(cuda(ref[None]), 8, want_bits=True)
    assert_bits_close(b2[0].cpu().numpy(), ref_bits, 32400)
    deg = DeShuffler(key=0).set_shape((8,))
    assert np.array_equal(deg.degenerate_counts(c2[0].cpu().numpy(), 32400), P8)
    assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), 32400), P8)
    # our own marked frame decodes to the same bits as the reference's on the sign-determined blocks
    det = mask[::8, ::8].reshape(-1)
    own = bits[0].cpu().numpy()
    _log("raw_bits_own_marked", (own[det] != ref_bits[det]).sum(), int(det.sum()))
    assert (own[det] != ref_bits[det]).sum() <= budget(32400, BITS_FRAC)
    # the sign-ambiguous blocks (~15 % of this frame) are excluded from the pixel comparison above; what the
    # reference defines for them is the quantised MAGNITUDE of C21 (dct_encoder.py:30-35), hence the decoded bit:
    # check it on every one of them.  The only other admissible outcome is "exact zero on one side, rounding
    # noise on the other" (np.sign(0) == 0 keeps a zero, noise gets +-step), which must be rare.
    dbg = enc.debug
    d = eng.debug_planes(cuda(nat), alpha=20, wm=wm)
    amb = np.abs(dbg["c21_pre"]) <= C21_TOL
    assert amb.sum() == n_amb
    mag_ok = np.abs(np.abs(d["c21_post"]) - np.abs(dbg["c21_post"])) <= 2e-3
    zero_vs_noise = (dbg["c21_pre"] == 0) != (d["c21_pre"] == 0)
    _log("ambiguous_magnitude", (amb & ~mag_ok & ~zero_vs_noise).sum(), n_amb)
    _log("ambiguous_zero_vs_noise", (amb & zero_vs_noise).sum(), n_amb)
    assert (amb & ~mag_ok & ~zero_vs_noise).sum() == 0
    assert (amb & zero_vs_noise).sum() <= budget(n_amb, 1e-2)
    assert (d["c21_post"][d["c21_pre"] == 0] == 0).all()           # the kernel's own exact zeros stay zero
    # the ambiguous blocks of our own marked frame decode to the bit the quantised magnitude carries
    nz = amb.reshape(-1) & ~zero_vs_noise.reshape(-1) & (dbg["c21_pre"].reshape(-1) != 0)
    _log("raw_bits_own_marked_ambiguous", (own[nz] != ref_bits[nz]).sum(), int(nz.sum()))
    assert (own[nz] != ref_bits[nz]).sum() <= budget(int(nz.sum()), 1e-3)
    # and the raw bit-error rate against the embedded watermark is the same as the reference's to within 0.1 %
    assert abs((own != wm.reshape(-1)).mean() - (ref_bits != wm.reshape(-1)).mean()) < 1e-3


def test_soft_decision_extension(eng):
    """Optional soft read-out (build extension): on clean frames it agrees with the hard decision
    and with a host evaluation of the same formula; under heavy noise, adding soft sums over the frames of a
    segment recovers at least as many segments as the reference's per-frame hard vote."""
    import torch
    from offmark import fingerprint as fp
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.vote import soft_vote, vote_segments
    from offmark.synthetic import synthetic_frames
    H, W, S, F = 240, 320, 8, 10
    N = H * W // 64
    frames = synthetic_frames(S * F, H, W, seed=8000)
    payloads = np.stack([fp.payload_for_segment(s + 1) for s in range(S)])
    wm = np.stack([orc.shuffle_generate(p, (N,), 0) for p in payloads])
    seg = np.repeat(np.arange(S), F)
    marked = eng.embed(frames, wm, wm_row=seg.astype(np.int32))
    deg = DeShuffler(key=0).set_shape((8,))
    soft = eng.detect_soft(marked, 8).cpu().numpy()
    # against the oracle's independent NumPy statement of the metric (oracle.soft_sums: the reference's masks, C21 and
    # step from DctDecoderOracle + the extension's last line).  C21/step differs from the oracle's by <= ~1e-6 relative,
    # i.e. < 3 fixed-point units per block; 150 blocks per position here
    host_marked = marked.cpu().numpy()
    for i in (0, 17, 79):
        ref_soft = orc.soft_sums(host_marked[i], 8, alpha=20)
        assert np.abs(soft[i] - ref_soft).max() <= 3 * (N // 8), (i, soft[i], ref_soft)
        assert np.array_equal(soft[i] > 0, ref_soft > 0)
    for L in (5, 300, 2049):                          # LDS histogram and global-atomic paths, non-power-of-two lengths
        sl = eng.detect_soft(marked[:2], L).cpu().numpy()
        for i in range(2):
            ref_l = orc.soft_sums(host_marked[i], L, alpha=20)
            assert np.abs(sl[i] - ref_l).max() <= 3 * (N // L + 1), L
    clean = soft_vote(soft, deg.payload_idx, seg)
    assert all(np.array_equal(clean[s], payloads[s]) for s in range(S))
    g = torch.Generator(device="cuda").manual_seed(3)
    noisy = (marked.float() + 9.0 * torch.randn(marked.shape, device="cuda", generator=g)).round().clamp(0, 255).to(torch.uint8)
    counts, _ = eng.detect(noisy, 8)
    hard = vote_segments(deg.degenerate_counts(counts.cpu().numpy(), N), seg)
    softv = soft_vote(eng.detect_soft(noisy, 8).cpu().numpy(), deg.payload_idx, seg)
    n_hard = sum(np.array_equal(hard[s][0], payloads[s]) for s in range(S))
    n_soft = sum(np.array_equal(softv[s], payloads[s]) for s in range(S))
    print(f"noise sigma 9: hard per-frame vote recovers {n_hard}/{S} segments, summed soft decision {n_soft}/{S}")
    assert n_soft >= n_hard


@pytest.mark.parametrize("tag,H,W", [("1080p", 1080, 1920), ("4k", 2160, 3840)])
def test_grayscale_image_payload_at_scale(eng, tag, H, W):
    """GrayScale / DeGrayScale (generator/grayscale.py:16-31, degenerator/de_grayscale.py:15-23; tests/test.py:31-40,75
    pairs them with image payloads) with the reference's own 480x270 image tests/media/wms/numbers.jpeg: L = 129 600
    bits, far beyond the LDS histogram (global-atomic count path), on full frames.  1080p: capacity 32 400 < L -- the
    generator warns and truncates, the degenerator's empty slices are nan and the decoded image is all zero, as in the
    reference; 4K: capacity == L, one block per bit.  Pinned by tests/golden/grayscale_numbers_digest.npz (digests of
    a run of the reference's own modules; OpenCV arithmetic parity-unpinned) through the oracle."""
    import warnings
    import torch
    from PIL import Image
    from offmark.degenerator.de_grayscale import DeGrayScale
    from offmark.generator.grayscale import GrayScale
    g = np.load(os.path.join(GOLDEN, "grayscale_numbers_digest.npz"))
    img = np.asarray(Image.open(os.path.join(GOLDEN, "numbers.jpeg")).convert("L"))
    key, L, N = int(g["key"]), img.size, H * W // 64
    frame = orc.synthetic_frame(H, W, int(g[tag + "_seed"]))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        wm = GrayScale(key=key).generate_wm(img, (1, N))
    assert bool(caught) == (L > N)
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(wm)
    ref = orc.mark_frame(frame, enc)
    ref_bits = orc.check_frame(ref, orc.DctDecoderOracle(alpha=20)).reshape(-1)
    marked, counts, bits = eng.embed_detect(cuda(frame[None]), wm, L=L, want_bits=True)
    ok = np.abs(enc.debug["c21_pre"]) > C21_TOL
    assert_pixels_close(marked[0].cpu().numpy(), ref, np.kron(ok, np.ones((8, 8), bool)))
    c2, b2 = eng.detect(cuda(ref[None]), L, want_bits=True)
    assert_bits_close(b2[0].cpu().numpy(), ref_bits, N)
    own = bits[0].cpu().numpy()
    assert np.array_equal(counts[0].cpu().numpy()[: min(L, N)], np.array([own[i::L].sum() for i in range(min(L, N))]))
    assert not counts[0, N:].any()
    deg = DeGrayScale(key=key).set_shape(img.shape)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = orc.degrayscale(ref_bits, img.shape, key)
        with np.errstate(all="ignore"):
            got_ref = deg.degenerate_counts(c2[0].cpu().numpy(), N)
            got_own = deg.degenerate_counts(counts[0].cpu().numpy(), N)
    digest = np.unpackbits(g[tag + "_degenerated_packed"])[:L].reshape(img.shape) * 255
    assert got_ref.shape == img.shape and got_ref.dtype == np.uint8
    # device epilogue of the same counts
    perm = torch.as_tensor(deg.payload_idx, dtype=torch.int32).cuda()
    dev = eng.payloads(c2, N, perm)[0].cpu().numpy().reshape(img.shape) * 255
    if tag == "1080p":
        assert not want.any() and not got_ref.any() and not got_own.any() and not dev.any() and not digest.any()
    else:
        # one block per bit: the decoded image is the raw bit plane un-permuted; a differing raw bit is a differing pixel
        assert (got_ref != want).sum() <= budget(N, BITS_FRAC) and (dev != got_ref).sum() == 0
        assert (got_ref != digest).mean() < 1e-3              # digest: the NEP-50 run of the reference modules
        assert np.mean((got_own > 0) == (img > 127)) > 0.999
        print(f"4K GrayScale: decoded image agrees with numbers.jpeg on {np.mean((got_own > 0) == (img > 127)):.5f} of its pixels")


def test_grayscale_content_where_every_block_is_sign_ambiguous(eng):
    """R = G = B video: U is float rounding noise around 0.5 everywhere, so EVERY block's C21 sign is undefined
    (see the module docstring).  The quantised magnitude, the decoded bits and the payload must still agree
    with the oracle; pixels may differ by the sign of the +-step pattern."""
    from offmark.degenerator.de_shuffler import DeShuffler
    g = orc.synthetic_frame(240, 320, 1001)[:, :, 1]
    frame = np.repeat(g[:, :, None], 3, axis=2)
    wm = orc.shuffle_generate(P8, (1, 1200), 0)
    dbg = oracle_embed_debug(frame, wm, 20)
    assert (np.abs(dbg["c21_pre"]) <= C21_TOL).mean() > 0.99
    d = eng.debug_planes(cuda(frame), alpha=20, wm=wm)
    both_zero = (dbg["c21_pre"] == 0) & (d["c21_pre"] == 0)
    mag_ok = np.abs(np.abs(d["c21_post"]) - np.abs(dbg["c21_post"])) <= 2e-3
    zero_vs_noise = ((dbg["c21_pre"] == 0) != (d["c21_pre"] == 0))
    _log("gray_zero_vs_noise", zero_vs_noise.sum(), 1200)
    assert (~(mag_ok | zero_vs_noise)).sum() == 0 and zero_vs_noise.sum() <= budget(1200, 2e-2)
    assert (d["c21_post"][both_zero] == 0).all()
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(wm)
    ref = orc.mark_frame(frame, enc)
    ref_bits = orc.check_frame(ref, orc.DctDecoderOracle(alpha=20)).reshape(-1)
    marked, counts, bits = eng.embed_detect(cuda(frame[None]), wm, L=8, want_bits=True)
    deg = DeShuffler(key=0).set_shape((8,))
    assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), 1200), P8)
    assert np.array_equal(orc.deshuffle(ref_bits, 8, 0), P8)
    _log("gray_raw_bits", (bits[0].cpu().numpy() != ref_bits).sum(), 1200)
    assert (bits[0].cpu().numpy() != ref_bits).sum() <= budget(1200, 2e-2)
    assert np.abs(marked[0].cpu().numpy().astype(int) - ref.astype(int)).max() <= 2 * 60     # at most a flipped +-step pattern


def _content(rng, H, W, kind):
    """Frame contents that steer the masks and the quantiser into their different branches."""
    if kind == "synthetic":
        return orc.synthetic_frame(H, W, int(rng.integers(1 << 30)))
    if kind == "noise":
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if kind == "flat":
        return np.broadcast_to(rng.integers(0, 256, 3, dtype=np.uint8), (H, W, 3)).copy()
    if kind == "dark":                                 # block means below 15 / 25: the luminance mask's dark steps
        return rng.integers(0, 40, (H, W, 3), dtype=np.uint8)
    if kind == "bright":                               # saturation: np.clip on the way back to u8
        return rng.integers(235, 256, (H, W, 3), dtype=np.uint8)
    if kind == "ramp":                                 # smooth gradients, exact zeros and tiny coefficients
        y, x = np.mgrid[0:H, 0:W]
        return np.stack([(x * 255 // max(W - 1, 1)), (y * 255 // max(H - 1, 1)), ((x + y) % 256)], -1).astype(np.uint8)
    if kind == "checker":                              # maximal texture energy
        y, x = np.mgrid[0:H, 0:W]
        v = (((x // 2) + (y // 3)) % 2 * 255).astype(np.uint8)
        return np.stack([v, 255 - v, v], -1)
    raise ValueError(kind)


def test_random_shapes_contents_and_strengths(eng):
    """Seeded sweep over frame sizes (multiples of 8 or not, 16-byte aligned rows or not), batch sizes, alphas,
    payload lengths and frame contents; every frame against the oracle, embed and detect."""
    rng = np.random.default_rng(4242)
    kinds = ["synthetic", "noise", "flat", "dark", "bright", "ramp", "checker"]
    for trial in range(28):
        H, W = int(rng.integers(8, 150)), int(rng.integers(8, 200))
        if trial % 4 == 0:
            W = (W // 16 + 1) * 16                      # aligned fast path
        n = int(rng.integers(1, 5))
        alpha = float(rng.choice([5.0, 10.0, 20.0, 33.5]))
        L = int(rng.choice([2, 5, 8, 13]))
        payload = rng.integers(0, 2, L)
        N, nblk = H * W // 64, (H // 8) * (W // 8)
        wm = orc.shuffle_generate(payload, (1, N), 3)
        frames = np.stack([_content(rng, H, W, kinds[(trial + k) % len(kinds)]) for k in range(n)])
        got, counts, bits = eng.embed_detect(cuda(frames), wm, L=L, alpha=alpha, want_bits=True)
        got = got.cpu().numpy()
        for k in range(n):
            enc = orc.DctEncoderOracle(alpha=alpha)
            enc.read_wm(wm)
            ref = orc.mark_frame(frames[k], enc)
            mask, _ = sign_determined_pixels(frames[k], wm, alpha)
            assert_pixels_close(got[k], ref, mask)
            ref_bits = orc.check_frame(ref, orc.DctDecoderOracle(alpha=alpha))
            _, b2 = eng.detect(cuda(ref[None]), L, alpha=alpha, want_bits=True)
            assert_bits_close(b2[0].cpu().numpy(), ref_bits, nblk)
        # the fused verify (bits of the frames the engine itself wrote) agrees with a separate detect call
        c3, b3 = eng.detect(cuda(got), L, alpha=alpha, want_bits=True)
        assert np.array_equal(c3.cpu().numpy(), counts.cpu().numpy())
        assert np.array_equal(b3.cpu().numpy(), bits.cpu().numpy())


def test_bench_dry_run_of_the_collective_path():
    """bench.py with a one-rank RCCL group: the process-group set-up, the side-stream all-gather, the barriers and
    the MAX all-reduce of the N>1 path all execute (on one GPU), and the line keeps its contract fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-collectives", "--frames", "8",
                        "--height", "240", "--width", "320", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:400]        # ONE line on stdout: RCCL's version banner goes to stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert line["payload_bit_exact"] and line["n_gpus"] == 1 and line["steps"] == 3
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["achieved"] > 0


@pytest.mark.parametrize("config,extra", [(4, []), (5, []), (5, ["--codec", "dwtdctsvd"]), (5, ["--codec", "dwtdctsvd", "--blk", "8"]),
                                          (3, ["--frames", "6", "--chunk", "4"]), (2, ["--frames", "24"])])
def test_bench_configs_run_at_one_gpu(config, extra):
    """bench.py --config 2/3/4/5 (BASELINE.json configs[1..4]) at N=1, shortened: the line keeps its contract, the
    payloads / votes / leak copy sequence check out, and the extras of the default config are present."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    small = ["--height", "240", "--width", "320"] if config != 2 else []
    if config in (4, 5):
        small += ["--frames", "6"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", str(config), "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", *small, *extra], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["payload_bit_exact"] and line["n_gpus"] == 1 and line["value"] > 0
    assert line["scaling"] == ("strong" if config in (4, 5) else "weak")
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    assert line["value_second_pass"] > 0 and line["second_pass"]["votes_ok"]
    if config in (2, 3):
        assert line["value_two_streams"] > 0 and line["two_streams"]["votes_ok"]
    assert line["hbm_copy_GBps"] > 1000 and line["hbm_read_GBps"] > 1000      # sanity only: a rate, not a ranking
    if config in (2, 3, 4):    # the fused mark kernel in both tile orders, interleaved in the same process
        mo = line["mark_order"]
        assert mo["xcd_ms"] > 0 and mo["linear_ms"] > 0 and mo["shipped"] in ("xcd", "linear") and mo["xcc_deal"]["xcds"] >= 1
        assert line["config"]["tile_order"] == mo["shipped"]          # what the TIMED region used
        assert mo["policy"] == line["config"]["tile_order_policy"] == "static rule"       # the default measures nothing
    if config == 5:        # BASELINE configs[4]: the attack suite is reported next to the line; clean and noisy leaks must resolve
        assert line["config"]["codec"] == ("dwtdctsvd" if "dwtdctsvd" in extra else "dct")      # --codec is honoured
        assert ("svd" in line["roofline"]["kernel"]) == ("dwtdctsvd" in extra)
        atk = line["attacks"]
        assert atk["none"]["copies_recovered"] and atk["none"]["payload_ber"] == 0 and atk["noise_sigma2"]["copies_recovered"]
        assert all(k in atk for k in ("scale_2_3_and_back", "crop16_and_resize_back", "jpeg_q95_420", "jpeg_q75_420"))
    if config in (2, 3):   # the two operations mark.py / detect.py perform, each alone (VERDICT r4 missing 2)
        eo, do = line["embed_only"], line["detect_only"]
        assert eo["value"] > 0 and 0 < eo["frac_of_peak"] < 1 and eo["frac_of_measured_copy"] > 0 and eo["algorithmic_bytes_per_frame"] % 6 == 0
        assert do["value"] > 0 and 0 < do["frac_of_peak"] < 1 and do["frac_of_measured_read"] > 0 and do["payload_ok"]
    if config == 2:
        assert line["value_separate_detect"] > 0 and line["separate_detect"]["votes_ok"]
        assert line["planar_i420"]["payload_ok"] and line["planar_i420"]["value"] > 0
        assert line["dwtdctsvd"]["payload_ok"] and line["dwtdctsvd"]["value"] > 0
        assert line["dwtdctsvd_blk8"]["payload_ok"] and line["dwtdctsvd_blk8"]["value"] > 0
        assert "device_under_load" in line                     # a sample or an error text, never a crash of the line
        py = line["plugin_yuv32f"]                             # the literal encode(yuv) / decode(yuv) boundary has a number
        assert py["dct"]["payload_ok"] and py["dwtdctsvd"]["payload_ok"] and py["dct"]["encode_decode_fps"] > 0
        # both rates are reported, not ranked: the PCIe leg shares the host with whatever else runs on the box
        assert line["pcie_inclusive"]["i420"]["frames_per_s"] > 0 and line["pcie_inclusive"]["rgb24"]["frames_per_s"] > 0


@pytest.mark.parametrize("config,size", [(2, ["--height", "360", "--width", "640", "--frames", "12"]), (4, ["--height", "240", "--width", "320", "--frames", "6"])])
def test_bench_line_checks_its_own_timed_frames_against_the_oracle(config, size):
    """the line itself compares frames of the TIMED batch and the marked frames the timed steps wrote with
    the C oracle's embed + detect (beside the CPU baseline): `oracle_check` with the parity tests' budgets, and a budget overrun
    turns payload_bit_exact false.  Also in every default line now: the non-fused mark kernel's own figures (`kernels.mark`),
    the contract read from idle (`value_no_preheat`) and CPU baseline variant A over ten frames.  dct_decoder.py:10-27."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", str(config), "--steps", "3", "--warmup", "1",
                        "--cpu-seconds", "1", *size], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    oc = line["oracle_check"]
    assert "error" not in oc, oc
    assert oc["within_budget"] and oc["payload_equal"] and line["payload_bit_exact"]
    assert oc["frames"] == len(oc["frame_indices"]) >= 2 and oc["max_pixel_difference"] <= 1 and oc["raw_bits_differing"] <= 1
    assert oc["pixels_compared"] > 0 and oc["pixels_differing_over_determined_blocks"] <= max(1, oc["pixels_compared"] // 100000)
    assert line["value_no_preheat"] > 0 and line["no_preheat"]["votes_ok"] and line["config"]["preheat_ms"] > 0
    assert line["config"]["placement_probe"]["candidates"] == 8                 # (these shortened batches are below the probe's size floor)
    if config == 2:
        k = line["kernels"]["mark"]
        assert k["launches"] >= 3 and k["avg_launch_ms"] > 0 and 0 < k["frac_of_peak"] < 1 and k["achieved_GBps"] > 0
        a_ = line["cpu_baseline"]["variants"]["A_reference_shaped_loop"]
        assert a_["frames"] == 10 and a_["payload_ok"] and a_["value"] > 0 and a_["cores"] == 1
        for key in ("dwtdctsvd", "dwtdctsvd_blk8"):      # the codec mark.py / detect.py construct: the same frames against the NumPy oracle
            so = line[key]["oracle_check"]
            assert so["within_budget"] and so["payload_equal"] and so["max_pixel_difference"] <= 1 and so["raw_bits_differing"] <= 1, (key, so)
            assert so["tiles"] > 0 and so["tiles_left_out"] < so["tiles"] // 2
    print("oracle_check:", {k_: v for k_, v in oc.items() if k_ != "note"})


def test_bench_config3_4k_at_its_stated_size():
    """BASELINE.json configs[2] (4K, HBM-bound stress) AT ITS STATED SIZE: 1000 frames of 3840x2160 per step = 24.9 GB in +
    24.9 GB out, three equal internal chunks (334 frames) under the default 8 GiB cap.  Payloads exact; the path must hold
    >= 0.55 of the 8 TB/s spec (measured 0.62, profiles/r3_bench_config3_4k_1000frames.json) in the timed region or in
    the pass right after it (a short timed region from idle sits on the clock ramp)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "3", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["payload_bit_exact"] and line["second_pass"]["votes_ok"] and line["config"]["frames_per_gpu"] == 1000
    assert line["config"]["chunk_frames"] * line["config"]["chunks_per_step"] >= 1000 > line["config"]["chunk_frames"] * (line["config"]["chunks_per_step"] - 1)
    assert line["config"]["chunks_per_step"] > 1                    # several, equal chunks
    spec = 8000.0
    timed = line["path"]["frac_of_peak"]
    after = line["value_second_pass"] * line["path"]["bytes_per_frame"] / 1e9 / spec
    print(f"config 3 at 1000 frames: path {timed:.3f} of spec in the timed region, {after:.3f} in the second pass, "
          f"{line['value']:.0f} frames/s, dominant kernel {line['roofline']['frac']:.3f}")
    assert max(timed, after) >= 0.55, (timed, after)


@pytest.mark.parametrize("config,extra", [(4, []), (5, []), (5, ["--codec", "dwtdctsvd"]), (2, ["--frames", "40"])])
def test_bench_emulates_rank_0_of_an_8_rank_job(config, extra):
    """bench.py --emulate-world 8: ONE GPU processes rank 0's shard of the 8-rank job (1 segment x 48 frames
    of configs 4/5; its own frames of config 2) but gathers and votes over all 8 ranks' payloads, times the whole job on the
    same GPU, and reports a predicted speed-up.  A 48-frame shard runs as a captured hipGraph with several steps per host
    iteration.  Checked: the contract, the votes of the shard run and of the whole-job run, and that the prediction is a number
    in the plausible range (the value itself is a measurement, recorded in profiles/, not a pass/fail bar here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", str(config), "--emulate-world", "8", "--steps", "20",
                        "--warmup", "5", "--no-cpu-baseline", *extra], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    em = line["emulation"]
    print(f"config {config} {extra}: emulated 8 ranks: shard {em['shard_ms_per_step']} ms/step, whole job {em['full_job_ms_per_step']} ms/step, "
          f"predicted speed-up {em['predicted_speedup']}, host {em['host_ms_per_step']}, group {em['steps_per_host_iteration']}, graph {em['hipgraph']}")
    assert line["payload_bit_exact"] and em["full_job_votes_ok"] and line["second_pass"]["votes_ok"]
    assert line["n_gpus"] == 1 and em["world"] == 8 and "EMULATED 8-RANK" in line["config"]["workload"]
    assert em["shard_frames"] == line["config"]["frames_per_gpu"] == (40 if config == 2 else 48)
    assert em["total_frames"] == 8 * em["shard_frames"]
    assert em["hipgraph"] and em["steps_per_host_iteration"] > 1 and 20 % em["steps_per_host_iteration"] == 0
    assert 1.0 < em["predicted_speedup"] <= 10.0          # above 8 is possible: a graphed shard overlaps its steps on two branches
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1       # from the event pass after the graphed region


@pytest.mark.parametrize("config,launcher", [(4, "driver"), (2, "self"), (4, "self")])
def test_bench_two_ranks_gloo_on_one_device(config, launcher):
    """The N>1 flow of bench.py (sharding of segments / frames over ranks, all-gather of the payloads, vote on every
    rank, MAX over ranks) with two processes on the one GPU of the box, gloo standing in for RCCL (a one-GPU box cannot
    run RCCL across ranks).  launcher "driver": started the way the driver starts it (torch.distributed.run around
    bench.py); "self": plain `python bench.py --gpus 2 ...`, bench.py spawns its own ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import socket
    with socket.socket() as sock:                  # a port that is free now (the parametrised runs follow each other)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    head = [sys.executable]
    if launcher == "driver":
        head += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(port)]
    cmd = head + [os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                  "--config", str(config), "--height", "240", "--width", "320", "--frames", "8", "--steps", "3", "--warmup", "1",
                  "--no-cpu-baseline"] + (["--no-extras"] if config != 2 else ["--side-measurements"])      # config 2: the side measurements run on every rank too
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(out_lines) == 1, r.stdout[-2000:]
    line = json.loads(out_lines[-1])
    assert line["n_gpus"] == 2 and line["payload_bit_exact"]
    assert line["collective"] == {"backend": "gloo", "ranks": 2, "self_launched": launcher == "self",
                                  "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}} and line["rccl_ranks"] is None      # both launch paths: the same rank environment
    # what every rank measured by itself: three figures per rank, and how far the slowest is from the median
    pr, sc = line["per_rank"], line["scaling_efficiency_inputs"]
    assert len(pr["ms_per_step"]) == len(pr["dominant_kernel_ms"]) == len(pr["analyze_ms"]) == 2
    assert all(x is not None and x > 0 for x in pr["ms_per_step"]) and max(pr["ms_per_step"]) <= line["ms_per_step"] + 1e-4       # (the line rounds to 4 decimals)
    assert all(x is not None and x > 0 for x in pr["analyze_ms"]) and pr["dominant_kernel"] == "mark_fused"
    assert sc["slowest_rank"] in (0, 1) and sc["slowest_over_median"] >= 1.0 and abs(sc["slowest_ms_per_step"] - max(pr["ms_per_step"])) < 1e-9
    assert line["config"]["frames_per_gpu"] == (8 if config == 2 else 4 * 8)      # config 4: 8 segments / 2 ranks x 8 frames
    assert line["config"]["steps_per_host_iteration"] == 3 and line["config"]["hipgraph"]   # small shards: the three steps are ONE graph, gathered and voted on together
    assert line["host_ms_per_step"]["over"] == "max over ranks" and "placement" in line
    if config == 2:
        assert line["value_second_pass"] > 0 and line["second_pass"]["votes_ok"] and line["separate_detect"]["votes_ok"]
        assert line["planar_i420"]["payload_ok"] and line["dwtdctsvd"]["payload_ok"]


def test_bench_rank_without_a_shard_stays_in_step_with_its_peers():
    """with more ranks than segments a rank has nothing to mark, but every step still gathers and the
    pre-heat ends on an all-reduce vote -- that rank used to return from the pre-heat at once and fall out of step with its
    peers' collectives (a hang).  Three ranks over a two-segment job (gloo, one device): rank 2 holds no shard; the job must end
    with a line, the votes right, and three entries per rank figure."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--backend", "gloo", "--single-device", "--config", "4",
                        "--segments", "2", "--height", "240", "--width", "320", "--frames", "8", "--steps", "4", "--warmup", "1",
                        "--preheat-ms", "40", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["payload_bit_exact"] and line["config"]["preheat_ms"] > 0
    assert line["config"]["frames_per_gpu"] == 8 and line["config"]["steps_per_host_iteration"] == 1       # ragged shards: 1 + 1 + 0 segments
    assert len(line["per_rank"]["ms_per_step"]) == 3 and line["per_rank"]["dominant_kernel_ms"][2] is None  # rank 2 launched nothing
    assert line["value_no_preheat"] > 0 and line["no_preheat"]["votes_ok"]


@pytest.mark.parametrize("where,flags", [("second_pass", []), ("mark_order", ["--side-measurements"]), ("timed", [])])
def test_bench_rank_failure_ends_the_whole_job(where, flags):
    """at N > 1 an exception in ONE rank (here injected into rank 1: in the timed steps, in the
    second pass, inside a side measurement that used to sit in a try/except) must end the WHOLE job quickly with a non-zero exit
    and no JSON line -- never leave the other rank in a barrier until somebody's time limit.  Default N > 1 runs also skip the
    side measurements ("mark_order" is not in their line)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = ["--backend", "gloo", "--single-device", "--height", "240", "--width", "320", "--frames", "8", "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline", *flags]
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", *args], capture_output=True, text=True, timeout=300,
                       env=dict(env, OFMK_BENCH_INJECT_FAILURE=f"1:{where}"), cwd=root)
    took = time.perf_counter() - t0
    assert r.returncode != 0 and took < 60, (r.returncode, took, r.stderr[-1500:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout[-500:]
    assert f"injected failure in rank 1 at {where}" in r.stderr
    if where == "second_pass":          # the healthy twin, and the one-GPU run of the same arguments, at a size where the work (not the interpreter start-up) sets the time
        big = ["--backend", "gloo", "--single-device", "--frames", "60", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"]
        t0 = time.perf_counter()
        ok2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", *big], capture_output=True, text=True,
                             timeout=600, env=env, cwd=root)
        t2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        ok1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", *big[3:]], capture_output=True, text=True,
                             timeout=600, env=env, cwd=root)
        t1 = time.perf_counter() - t0
        assert ok2.returncode == 0 and ok1.returncode == 0, (ok2.stderr[-1500:], ok1.stderr[-1500:])
        line2 = [l for l in ok2.stdout.splitlines() if l.startswith("{")]
        assert len(line2) == 1 and "mark_order" not in line2[0] and "second_pass" in line2[0]       # N > 1 default: value + second pass only
        print(f"bench wall time: N=2 (gloo, one device) {t2:.1f} s, N=1 {t1:.1f} s")        # a figure, not a gate: whole-process times on a shared box


def test_c_host_program_over_the_abi(tmp_path):
    """examples/abi_demo.c: a plain C process (no Python, no torch in it) allocates with the HIP runtime, marks and verifies
    with both codecs through the C ABI, reads the written frames with the stand-alone detectors and gets every payload
    back; an invalid call comes back as an error code with a text."""
    import subprocess
    from test_abi_and_host import build_abi_demo
    exe = str(tmp_path / "abi_demo")
    build_abi_demo(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.strip().endswith("abi_demo OK") and r.stdout.count("6 of 6 frames carry the payload") == 6
    assert "refused with code -1" in r.stdout
