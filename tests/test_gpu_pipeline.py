"""The plugin-level frame loop (offmark.video.embedder.Embedder / extractor.Extractor; reference
src/offmark/video/embedder.py:18-31, extractor.py:18-28) as a three-stream pipeline (offmark/video/pipeline.py).

A pipeline must not change a single byte: every form of hand-over (page-locked reader, page-locked writer block,
staging copies, per-frame read()/write() of duck-typed objects, rgb24 and 4:2:0 planes, mixed formats) is compared
BIT FOR BIT with one direct batch call of the engine on the same frames -- which the parity tests
(tests/test_gpu_parity.py, tests/test_gpu_planar.py) compare with the oracle.  Order of frames, ragged last batch,
more batches than pipeline slots, empty streams and a reader that fails mid-stream are covered.
"""
import numpy as np
import pytest

import offmark_oracle as orc

pytestmark = pytest.mark.gpu
P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
H, W = 64, 96


@pytest.fixture(scope="module")
def eng():
    import torch
    from offmark.engine import DctEngine
    torch.cuda.set_device(0)
    return DctEngine()


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def frames_rgb(n, h=H, w=W, seed=500):
    return np.stack([orc.synthetic_frame(h, w, seed + i) for i in range(n)])


def make_codec(codec, h=H, w=W):
    from offmark.generator.shuffler import Shuffler
    if codec == "dct":
        from offmark.embed.dct_encoder import DctEncoder as Enc
        from offmark.extract.dct_decoder import DctDecoder as Dec
    else:
        from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder as Enc
        from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder as Dec
    enc = Enc()
    enc.read_wm(Shuffler(key=0).generate_wm(P8, enc.wm_capacity((h, w, 3))))
    return enc, Dec()


class BareReader:
    """Only what the reference's FrameReader promises: read() and close() -- no size, no batches."""

    def __init__(self, frames, fail_at=None):
        self.frames, self.pos, self.fail_at, self.closed = frames, 0, fail_at, False

    def read(self):
        if self.fail_at is not None and self.pos == self.fail_at:
            raise IOError("decoder died")
        if self.pos >= len(self.frames):
            return None
        self.pos += 1
        return self.frames[self.pos - 1]

    def close(self):
        self.closed = True


class BareWriter:
    def __init__(self):
        self.frames, self.closed = [], False

    def write(self, frame):
        self.frames.append(np.array(frame))           # the pipeline hands out views of its staging buffer

    def close(self):
        self.closed = True


@pytest.mark.parametrize("codec", ["dct", "dwtdctsvd"])
@pytest.mark.parametrize("form", ["pinned", "registered", "pageable", "bare"])
def test_embedder_pipeline_equals_one_direct_batch_call(eng, codec, form):
    from offmark.video.embedder import Embedder
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark.video.frame_writer import ArrayFrameWriter
    from offmark.video.pipeline import pinned_empty
    n, B = 23, 5                                     # 5 batches (more than the 2 slots), ragged last one
    src = frames_rgb(n)
    enc, _ = make_codec(codec)
    want = enc.encode_frames_u8(cuda(src)).cpu().numpy()
    if form == "pinned":
        host = pinned_empty(src.shape)
        host[:] = src
        r, w = ArrayFrameReader(host, pin="already"), ArrayFrameWriter(capacity=n, frame_shape=src.shape[1:])
    elif form == "registered":
        r, w = ArrayFrameReader(src.copy(), pin=True), ArrayFrameWriter(capacity=n - 7, frame_shape=src.shape[1:])   # block too small: falls back
    elif form == "pageable":
        r, w = ArrayFrameReader(src), ArrayFrameWriter()
    else:
        r, w = BareReader(list(src)), BareWriter()
    emb = Embedder(r, enc, w, batch_frames=B)
    emb.start()
    assert emb.frames_marked == n and len(w.frames) == n and r.closed and w.closed
    assert np.array_equal(np.stack(w.frames), want)
    if form in ("pinned", "registered", "pageable"):     # array() is the whole stream, also when the block was too small
        assert np.array_equal(w.array(), want)


@pytest.mark.parametrize("layout,fmt", [("i420", "yuv420p"), ("nv12", "nv12")])
def test_planar_readers_and_writers(eng, layout, fmt):
    """yuv420p / nv12 frames [H*3/2, W] through Embedder and Extractor == the engine's planar batch calls; mixed
    reader / writer formats == the explicit conversion chain."""
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.video.embedder import Embedder
    from offmark.video.extractor import Extractor
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark.video.frame_writer import ArrayFrameWriter
    n, B, h, w = 11, 4, 240, 320
    rgb = frames_rgb(n, h, w, seed=1001)
    enc, dec = make_codec("dct", h, w)
    planes = eng.rgb_to_yuv420(cuda(rgb), layout)
    want = eng.embed_yuv420(planes, h, w, enc._device_wm(h * w // 64), layout=layout).cpu().numpy()
    src = planes.cpu().numpy().reshape(n, h * 3 // 2, w)
    wr = ArrayFrameWriter(pix_fmt=fmt)
    Embedder(ArrayFrameReader(src, pix_fmt=fmt), enc, wr, batch_frames=B).start()
    got = np.stack(wr.frames)
    assert got.shape == src.shape and np.array_equal(got.reshape(n, -1), want)
    ex = Extractor(ArrayFrameReader(got, pix_fmt=fmt), dec, DeShuffler(key=0).set_shape(P8.shape), batch_frames=B)
    ex.start()
    assert len(ex.patterns) == n and all(np.array_equal(p, P8) for p in ex.patterns)      # survives its own 4:2:0 subsampling
    # rgb24 in, planes out: mark in RGB, convert on the device
    wr2 = ArrayFrameWriter(pix_fmt=fmt)
    Embedder(ArrayFrameReader(rgb), enc, wr2, batch_frames=B).start()
    chain = eng.rgb_to_yuv420(enc.encode_frames_u8(cuda(rgb)), layout).cpu().numpy()
    assert np.array_equal(np.stack(wr2.frames).reshape(n, -1), chain)
    # planes in, rgb24 out
    wr3 = ArrayFrameWriter()
    Embedder(ArrayFrameReader(src, pix_fmt=fmt), enc, wr3, batch_frames=B).start()
    chain3 = enc.encode_frames_u8(eng.yuv420_to_rgb(planes, h, w, layout)).cpu().numpy()
    assert np.array_equal(np.stack(wr3.frames), chain3)
    # a codec without planar kernels (DwtDctSvd) takes the conversion route on the device
    enc_s, dec_s = make_codec("dwtdctsvd", h, w)
    wr4 = ArrayFrameWriter(pix_fmt=fmt)
    Embedder(ArrayFrameReader(src, pix_fmt=fmt), enc_s, wr4, batch_frames=B).start()
    chain4 = eng.rgb_to_yuv420(enc_s.encode_frames_u8(eng.yuv420_to_rgb(planes, h, w, layout)), layout).cpu().numpy()
    assert np.array_equal(np.stack(wr4.frames).reshape(n, -1), chain4)
    ex4 = Extractor(ArrayFrameReader(np.stack(wr4.frames), pix_fmt=fmt), dec_s, DeShuffler(key=0).set_shape(P8.shape), batch_frames=B)
    ex4.start()
    want_counts, _ = dec_s.decode_frames_u8(eng.yuv420_to_rgb(cuda(chain4), h, w, layout), 8)
    want_payloads = DeShuffler(key=0).set_shape(P8.shape).degenerate_counts(want_counts.cpu().numpy(), h * w // 64)
    assert np.array_equal(np.stack(ex4.patterns), want_payloads)


@pytest.mark.parametrize("codec", ["dct", "dwtdctsvd"])
def test_extractor_pipeline_equals_direct_detect(eng, codec):
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.video.extractor import Extractor
    from offmark.video.frame_reader import ArrayFrameReader
    n, B = 19, 4
    enc, dec = make_codec(codec)
    marked = enc.encode_frames_u8(cuda(frames_rgb(n))).cpu().numpy()
    counts, _ = dec.decode_frames_u8(cuda(marked), 8)
    want = DeShuffler(key=0).set_shape(P8.shape).degenerate_counts(counts.cpu().numpy(), H * W // 64)
    for reader in (ArrayFrameReader(marked), ArrayFrameReader(marked.copy(), pin=True), BareReader(list(marked))):
        ex = Extractor(reader, dec, DeShuffler(key=0).set_shape(P8.shape), batch_frames=B)
        ex.start()
        assert reader.closed and np.array_equal(np.stack(ex.patterns), want)
        assert np.array_equal(ex.most_common()[0], P8)


@pytest.mark.parametrize("L", [8, 5, 30])
def test_extractor_and_copy_marking_with_blk8_divide_by_the_decoders_own_bit_count(eng, L):
    """DwtDctSvdDecoder(blk=8) returns row*col//256 bits (dwt_dct_svd_decoder.py:14), a quarter of the DCT codec's
    row*col//64, and DeShuffler's means are taken over the slices of THAT vector (de_shuffler.py:17-18).  The batched Extractor
    (counts from the kernel + degenerate_counts) must give, frame by frame, what the reference's per-frame route gives:
    decode(yuv) -> degenerate(bits) -- also when L does not divide the bit count (slice lengths differ) and when the frame has
    fewer tiles than payload positions (64x96: 24 tiles < L = 30: empty slices are nan upstream, the payload decodes to zeros)."""
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder
    from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder
    from offmark.generator.shuffler import Shuffler
    from offmark.video.color import bgr2yuv
    from offmark.video.extractor import Extractor
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark import fingerprint as fp
    n = 7
    payload = np.random.default_rng(L).integers(0, 2, L)
    payload[:2] = (0, 1)                                      # not constant: the mid-range threshold needs both values
    enc, dec = DwtDctSvdEncoder(blk=8), DwtDctSvdDecoder(blk=8)
    enc.read_wm(Shuffler(key=3).generate_wm(payload, enc.wm_capacity((H, W, 3))))
    assert dec.bits_per_frame(H, W) == H * W // 256 == 24
    marked = enc.encode_frames_u8(cuda(frames_rgb(n))).cpu().numpy()
    deg = DeShuffler(key=3).set_shape(payload.shape)
    with np.errstate(all="ignore"):
        per_frame = np.stack([deg.degenerate(dec.decode(bgr2yuv(f.astype(np.float32)))) for f in marked])     # the reference's route
    ex = Extractor(ArrayFrameReader(marked), dec, DeShuffler(key=3).set_shape(payload.shape), batch_frames=3)
    ex.start()
    assert np.array_equal(np.stack(ex.patterns), per_frame)
    if L == 30:
        assert not per_frame.any()                            # nan threshold upstream: nothing compares greater
    elif L == 8:
        assert np.array_equal(per_frame, np.tile(payload, (n, 1)))
    if L == 8:      # mark_segment_copies verifies its copies with the same rule
        copies, side = fp.mark_segment_copies(DwtDctSvdEncoder(blk=8), DwtDctSvdDecoder(blk=8), cuda(frames_rgb(8, 128, 192)),
                                              np.repeat([1, 2], 4), num_copies=2)
        assert side["failed_segments"] == [] and len(copies) == 2


def test_empty_stream_and_failing_reader(eng):
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.video.embedder import Embedder
    from offmark.video.extractor import Extractor
    enc, dec = make_codec("dct")
    r, w = BareReader([]), BareWriter()
    emb = Embedder(r, enc, w, batch_frames=4)
    emb.start()
    assert emb.frames_marked == 0 and w.frames == [] and r.closed and w.closed
    ex = Extractor(BareReader([]), dec, DeShuffler(key=0).set_shape(P8.shape))
    ex.start()
    assert ex.patterns == [] and ex.most_common() == (None, None)
    # the reader dies at its 10th read of 14 frames: the error reaches the caller (no hang, no silent truncation); every
    # frame read before it (9 of them: two full batches and one frame of the third) is marked and written, in order --
    # what the reference's one-frame-at-a-time loop would have done before the exception
    src = frames_rgb(14)
    w2 = BareWriter()
    emb2 = Embedder(BareReader(list(src), fail_at=9), enc, w2, batch_frames=4)
    with pytest.raises(IOError, match="decoder died"):
        emb2.start()
    want = enc.encode_frames_u8(cuda(src[:9])).cpu().numpy()
    assert emb2.frames_marked == len(w2.frames) == 9
    assert np.array_equal(np.stack(w2.frames), want)
    # a frame of the wrong size in the middle of the stream is refused, not read out of bounds
    bad = list(src[:6]) + [np.zeros((H + 8, W, 3), np.uint8)]
    with pytest.raises(ValueError, match="expected"):
        Embedder(BareReader(bad), enc, BareWriter(), batch_frames=4).start()


def test_engine_refuses_a_bad_destination(eng):
    """`out` goes to the kernels as a raw pointer, so it is checked first."""
    import torch
    f = cuda(frames_rgb(2))
    wm = np.zeros((1, H * W // 64), np.uint8)
    for bad in (torch.empty_like(f).cpu(), torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda"),
                torch.empty((2, H, W, 3), dtype=torch.float32, device="cuda"),
                torch.empty((2, H, W, 6), dtype=torch.uint8, device="cuda")[..., ::2]):
        for call in (lambda o: eng.embed(f, wm, out=o), lambda o: eng.svd_embed(f, wm, out=o),
                     lambda o: eng.embed_detect(f, wm, 8, out=o)):
            with pytest.raises(ValueError, match="out must be"):
                call(bad)
    planes = eng.rgb_to_yuv420(f)
    with pytest.raises(ValueError, match="out must be"):
        eng.embed_yuv420(planes, H, W, wm, out=torch.empty_like(planes).cpu())
    with pytest.raises(ValueError, match="unknown 4:2:0 layout"):
        eng.embed_yuv420(planes, H, W, wm, layout="yv12")


def test_out_of_range_device_row_map_is_clamped_never_read_out_of_bounds(eng):
    """the per-frame watermark-row map is this build's extension (the reference has one watermark per
    encoder, dct_encoder.py:10-11), so its safety is too.  A DEVICE-resident map is not inspected on the host (that would
    synchronise); the kernels clamp every entry into [0, n_wm): negative and too-large entries read the last row.  The
    table is the last allocation made here and the bad entries point megabytes away from it -- an unclamped read would
    fault or mark garbage.  With debug checks on, the host refuses the map."""
    import torch
    from offmark.engine import DctEngine
    f = cuda(frames_rgb(4))
    N = H * W // 64
    rng = np.random.default_rng(3)
    table = cuda(rng.integers(0, 2, (2, N), dtype=np.uint8))
    bad = torch.tensor([-1, 1 << 20, 1, -(1 << 30)], dtype=torch.int32, device="cuda")
    clamped = np.array([1, 1, 1, 1], np.int32)
    for call in (lambda r: eng.embed(f, table, wm_row=r),
                 lambda r: eng.svd_embed(f, table, wm_row=r),
                 lambda r: eng.embed_detect(f, table, 8, wm_row=r)[0],
                 lambda r: eng.svd_embed_detect(f, table, 8, wm_row=r)[0],
                 lambda r: eng.embed_yuv420(eng.rgb_to_yuv420(f), H, W, table, wm_row=r),
                 lambda r: eng.encode_yuv(f.float(), table, wm_row=r)):
        got, want = call(bad), call(clamped)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    with pytest.raises(ValueError, match="wm_row entries"):
        eng.embed(f, table, wm_row=bad.cpu().numpy())                 # host maps are always checked
    strict = DctEngine()
    strict.debug_checks = True
    with pytest.raises(ValueError, match="wm_row entries"):
        strict.embed(f, table, wm_row=bad)
    assert torch.equal(strict.embed(f, table, wm_row=torch.tensor([0, 1, 1, 0], dtype=torch.int32, device="cuda")),
                       eng.embed(f, table, wm_row=[0, 1, 1, 0]))


def test_timing_options_do_not_dangle(eng):
    """an Opts made by Timing.opts() keeps the pool alive and is disarmed by Timing.close()."""
    import gc
    from offmark import _hip
    from offmark.engine import DctEngine
    f = cuda(frames_rgb(2))
    wm = np.zeros((1, H * W // 64), np.uint8)
    timing = _hip.Timing(8)
    e = DctEngine(opts=timing.opts())
    e.embed(f, wm)
    assert timing.collect()["analyze"]["launches"] == 1
    timing.close()                                # the engine still holds the options: they no longer name the pool
    assert not e.opts.timing
    e.embed(f, wm)
    e2 = DctEngine(opts=_hip.Timing(8).opts())    # the Timing object itself is dropped here ...
    gc.collect()
    e2.embed(f, wm)                               # ... but lives on through the options
    assert e2.opts._timing.collect()["mark"]["launches"] == 1
    import torch
    torch.cuda.synchronize()
    # the float32-YUV DwtDctSvd entry points honour the timing object too (both block sizes), and the planar kernels
    # report under kinds of their own
    t3 = _hip.Timing(16)
    e3 = DctEngine(opts=t3.opts())
    yuv = torch.rand((1, H, W, 3), device="cuda") * 255
    for blk in (4, 8):
        e3.svd_encode_yuv(yuv, wm, scale=15, blk=blk)
        e3.svd_decode_yuv(yuv, scale=15, blk=blk)
    got = t3.collect()
    assert got["svd"]["launches"] == 4 and got["svd"]["ms_total"] > 0
    planes = e3.rgb_to_yuv420(f)
    e3.embed_detect_yuv420(planes, H, W, wm, 8)
    got = t3.collect()
    assert got["planar_analyze"]["launches"] == 1 and got["planar_mark"]["launches"] == 1 and got["finalize"]["launches"] == 1
    assert [k for _, k in t3.durations()] == []                       # collect() rewound the pool
    e3.detect_yuv420(planes, H, W, 8)
    assert [k for _, k in t3.durations()] == ["planar_analyze", "finalize"]     # launch order, kinds by name
    t3.collect()
    # blk = 8 on a frame with no 16x16 tile launches nothing -- and must not take an event pair either (a pair taken for
    # a launch that never happens reports a stale duration after the pool's next rewind)
    tiny = torch.rand((1, 12, 12, 3), device="cuda") * 255          # ((12 / 4) * 2) / 8 = 0 tiles of 16x16, one 8x8 tile
    wm_tiny = np.zeros((1, 12 * 12 // 64), np.uint8)
    e3.svd_encode_yuv(tiny, wm_tiny, scale=15, blk=8)                 # (its decode has H*W//256 = 0 bits to return: nothing to call)
    assert t3.durations() == [] and t3.collect()["svd"]["launches"] == 0
    e3.svd_encode_yuv(tiny, wm_tiny, scale=15, blk=4)                 # blk = 4 has 8x8 tiles there: one launch, one pair
    assert t3.collect()["svd"]["launches"] == 1
    t3.close()


@pytest.mark.parametrize("pix_fmt", ["rgb24", "yuv420p"])
def test_mark_and_detect_drivers_over_file_readers_and_writers(eng, fake_ffmpeg, tmp_path, pix_fmt):
    """The call sequence of tests/mark.py:18-40 and tests/detect.py:17-31 -- FileDecoder -> Embedder -> FileEncoder, then
    FileDecoder -> Extractor -- with the ffmpeg child processes replaced by the test double of tests/conftest.py (no ffmpeg on
    the boxes; the shape of the bundled clip: 320x240, 209 frames).  The pipe is read straight into the pipeline's page-locked
    staging (read_batch_into); what lands in the written "file" equals one direct batch call, every frame decodes to the
    payload.  yuv420p: the planes travel instead of rgb24 (the reference's open question, frame_reader.py:27)."""
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder
    from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder
    from offmark.generator.shuffler import Shuffler
    from offmark.video.embedder import Embedder
    from offmark.video.extractor import Extractor
    from offmark.video.frame_reader import FileDecoder
    from offmark.video.frame_writer import FileEncoder
    h, w, n = 240, 320, 209
    base = [orc.synthetic_frame(h, w, 1001 + i) for i in range(11)]
    rgb = np.stack([np.roll(base[i % 11], 8 * (i // 11), axis=1) for i in range(n)])
    clip = rgb if pix_fmt == "rgb24" else eng.rgb_to_yuv420(cuda(rgb)).cpu().numpy().reshape(n, h * 3 // 2, w)
    in_file, out_file = fake_ffmpeg(tmp_path / "in.raw", clip, pix_fmt), str(tmp_path / "marked.raw")
    payload = np.array([0, 1, 1, 0, 0, 1, 0, 1])
    r = FileDecoder(in_file, pix_fmt=pix_fmt)
    wr = FileEncoder(out_file, r.width, r.height, pix_fmt=pix_fmt)
    frame_embedder = DwtDctSvdEncoder()
    capacity = frame_embedder.wm_capacity((r.height, r.width, 3))
    wm = Shuffler(key=0).generate_wm(payload, capacity)
    frame_embedder.read_wm(wm)
    video_embedder = Embedder(r, frame_embedder, wr)
    video_embedder.start()
    assert video_embedder.frames_marked == n
    rd = FileDecoder(out_file, pix_fmt=pix_fmt)
    written = rd.read_batch(n + 5)
    rd.close()
    if pix_fmt == "rgb24":
        want = frame_embedder.encode_frames_u8(cuda(rgb)).cpu().numpy()
    else:
        planes = cuda(clip.reshape(n, -1))
        want = eng.rgb_to_yuv420(frame_embedder.encode_frames_u8(eng.yuv420_to_rgb(planes, h, w))).cpu().numpy().reshape(clip.shape)
    assert written.shape == clip.shape and np.array_equal(written, want)
    degenerator = DeShuffler(key=0)
    degenerator.set_shape(payload.shape)
    video_extractor = Extractor(FileDecoder(out_file, pix_fmt=pix_fmt), DwtDctSvdDecoder(), degenerator)
    video_extractor.start()
    assert len(video_extractor.patterns) == n and all(np.array_equal(p, payload) for p in video_extractor.patterns)


def test_4k_frames_run_in_byte_bounded_batches(eng):
    """64 frames of 4K per batch would pin 14 GB of host memory; the pipeline cuts a batch to 512 MiB (21 frames of 4K rgb24).
    25 frames -> batches of 21 + 4, bit-equal to one direct call."""
    from offmark.video.embedder import Embedder
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark.video.frame_writer import ArrayFrameWriter
    from offmark.video.pipeline import batch_size, frame_shape
    h, w, n = 2160, 3840, 25
    assert batch_size(64, frame_shape("rgb24", h, w)) == 21
    rng = np.random.default_rng(12)
    small = rng.integers(0, 256, (n, h // 8, w // 8, 3), dtype=np.uint8)
    src = np.ascontiguousarray(np.repeat(np.repeat(small, 8, axis=1), 8, axis=2))       # blocky 4K frames, cheap to make
    src[:, ::3, ::5] ^= 0x15                                                           # plus some texture
    enc, _ = make_codec("dct", h, w)
    want = enc.encode_frames_u8(cuda(src)).cpu().numpy()
    wr = ArrayFrameWriter()
    emb = Embedder(ArrayFrameReader(src), enc, wr)                                     # default batch_frames = 64
    emb.start()
    assert emb.frames_marked == n and np.array_equal(np.stack(wr.frames), want)
