"""CPU-only checks: the C-ABI library loads and exports everything include/offmark_hip.h declares,
the host-side codecs match the reference's numpy-only modules (golden vectors), the pipeline
plumbing works with an in-memory reader/writer, and the product path refuses to run without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

import offmark_oracle as orc
from conftest import GOLDEN, PKG, ROOT

P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])


def header_symbols():
    text = open(os.path.join(ROOT, "include", "offmark_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ofmk_\w+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()                                   # hipcc cross-compiles gfx950 without a GPU
    from offmark import _hip
    lib = _hip.load()
    declared = header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in offmark_hip.h but not exported"
    assert sorted(_hip.SYMBOLS) == declared
    # the library keeps no mutable globals besides the per-thread error text (header: "re-entrant")
    nm = subprocess.run(["nm", "-C", _hip.lib_path()], capture_output=True, text=True).stdout
    writable = [l for l in nm.splitlines() if len(l.split()) >= 3 and l.split()[1] in "bBdD"
                and ("ofmk" in l or " g_" in l) and "(" not in l          # "name(args)" = a kernel's launch handle
                and "g_err" not in l and "guard variable" not in l]
    assert not writable, writable
    assert lib.ofmk_version() == _hip.ABI_VERSION == 6
    # pure host-side entry points are safe to call without a GPU
    assert lib.ofmk_workspace_bytes(1, 1080, 1920) == 32400 * 4 * 4 + 2 * 127 * 8 + 4096      # records + delta, 2 x 127 per-tile partial sums
    assert lib.ofmk_workspace_bytes(0, 1080, 1920) == 0 and lib.ofmk_workspace_bytes(1, 4, 1920) == 0


def test_code_object_targets_gfx950_only():
    from offmark import _hip
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _hip.lib_path()], capture_output=True, text=True)
    data = open(_hip.lib_path(), "rb").read()
    assert b"gfx950" in data and b"gfx942" not in data and b"gfx90a" not in data


@pytest.mark.skipif("__import__('torch').cuda.is_available()")
def test_product_path_fails_loudly_without_gpu():
    from offmark import _hip
    from offmark.embed.dct_encoder import DctEncoder
    from offmark.extract.dct_decoder import DctDecoder
    enc = DctEncoder()
    enc.read_wm(np.zeros((1, 1200)))
    with pytest.raises(_hip.HipError, match="no CPU fallback"):
        enc.encode(np.zeros((240, 320, 3), np.float32))
    with pytest.raises(_hip.HipError):
        DctDecoder().decode(np.zeros((240, 320, 3), np.float32))


def test_product_package_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(import|from)\s+offmark_oracle|oracle/", text, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_generators_and_degenerators_match_reference_modules():
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.generator.shuffler import Shuffler
    g = np.load(os.path.join(GOLDEN, "payload_codecs.npz"))
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files})
    for t in tags:
        key = int(t.split("_")[0][1:])
        p, wm, noisy, back = (g[t + s] for s in ("_payload", "_wm", "_noisy", "_back"))
        gen = Shuffler(key=key)
        assert gen.wm_type() == "bits"
        got = gen.generate_wm(p, wm.shape)
        assert got.shape == wm.shape and np.array_equal(got, wm)
        deg = DeShuffler(key=key).set_shape(p.shape)
        assert np.array_equal(deg.degenerate(noisy), back)
        counts = np.array([noisy.reshape(-1)[i::p.size].sum() for i in range(p.size)]).astype(np.int64)
        assert np.array_equal(deg.degenerate_counts(counts, noisy.size), back)


def test_grayscale_codecs_match_golden():
    from offmark.degenerator.de_grayscale import DeGrayScale
    from offmark.generator.grayscale import GrayScale
    g = np.load(os.path.join(GOLDEN, "frame63_crop_qr_k0_a20.npz"))
    qr = g["payload"]
    gen = GrayScale(key=0)
    assert gen.wm_type() == "grayscale"
    assert np.array_equal(gen.generate_wm(qr, g["wm"].shape), g["wm"])
    deg = DeGrayScale(key=0).set_shape(qr.shape)
    out = deg.degenerate(g["raw_bits"])
    assert out.shape == qr.shape and np.array_equal(out, g["degenerated"])
    bits = g["raw_bits"].reshape(-1)
    counts = np.array([bits[i::qr.size].sum() for i in range(qr.size)])
    assert np.array_equal(deg.degenerate_counts(counts, bits.size), g["degenerated"])
    with pytest.warns(UserWarning):
        GrayScale(key=0).generate_wm(qr, (1, 100))


def test_degenerate_edge_cases_match_reference_semantics():
    from offmark.degenerator.de_shuffler import DeShuffler
    deg = DeShuffler(key=0).set_shape((8,))
    # constant payload decodes to zeros (strict > at the mid-range threshold)
    assert deg.degenerate(np.ones(64)).tolist() == [0] * 8
    # fewer bits than payload positions: mean of an empty slice is nan -> nothing is above threshold
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = orc.deshuffle(np.array([1.0, 0, 1, 1, 0, 1]), 8, 0)
    assert np.array_equal(deg.degenerate(np.array([1.0, 0, 1, 1, 0, 1])), ref)
    assert np.array_equal(deg.degenerate_counts(np.array([1, 0, 1, 1, 0, 1, 0, 0]), 6), ref)


def test_pipeline_plumbing_with_in_memory_reader_writer():
    """Config 1 shape (mark.py + detect.py logic) on CPU: the generic per-frame path with the oracle's
    encoder/decoder standing in for the GPU codec; the Embedder/Extractor/reader/writer are the product's."""
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.generator.shuffler import Shuffler
    from offmark.video.embedder import Embedder
    from offmark.video.extractor import Extractor
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark.video.frame_writer import ArrayFrameWriter
    frames = np.stack([orc.synthetic_frame(240, 320, 1001 + i) for i in range(6)])
    r, w = ArrayFrameReader(frames), ArrayFrameWriter()
    assert (r.width, r.height) == (320, 240)
    enc = orc.DctEncoderOracle(alpha=20)
    capacity = enc.wm_capacity((r.height, r.width, 3))
    enc.read_wm(Shuffler(key=0).generate_wm(P8, capacity))
    Embedder(r, enc, w).start()
    assert r.closed and w.closed and len(w.frames) == 6 and w.frames[0].dtype == np.uint8
    assert np.array_equal(w.frames[0], orc.mark_frame(frames[0], enc))
    ex = Extractor(ArrayFrameReader(w.frames), orc.DctDecoderOracle(alpha=20), DeShuffler(key=0).set_shape(P8.shape))
    ex.start()
    assert len(ex.patterns) == 6 and all(np.array_equal(p, P8) for p in ex.patterns)
    best, freq = ex.most_common()
    assert np.array_equal(best, P8) and freq == 1.0


def test_extractor_accepts_duck_typed_reader_without_read_batch():
    """The reference's plugin API is duck typing (frame_reader.py:13 "TODO extend ABC"): a reader with only
    read()/close() must work with the per-frame path."""
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.video.extractor import Extractor

    class Bare:
        def __init__(self, frames):
            self.frames, self.i, self.closed = frames, 0, False

        def read(self):
            self.i += 1
            return self.frames[self.i - 1] if self.i <= len(self.frames) else None

        def close(self):
            self.closed = True

    frame = orc.synthetic_frame(64, 96, 2)
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(orc.shuffle_generate(P8, (1, 96), 0))
    marked = orc.mark_frame(frame, enc)
    r = Bare([marked, marked])
    ex = Extractor(r, orc.DctDecoderOracle(alpha=20), DeShuffler(key=0).set_shape(P8.shape))
    ex.start()
    assert r.closed and len(ex.patterns) == 2 and np.array_equal(ex.patterns[0], P8)


def test_file_decoder_is_gated_on_ffmpeg():
    import shutil
    from offmark.video.frame_reader import FileDecoder
    from offmark.video.frame_writer import FileEncoder
    if shutil.which("ffmpeg") is None:
        with pytest.raises(RuntimeError, match="ffmpeg"):
            FileDecoder("nope.mp4")
        with pytest.raises(RuntimeError, match="ffmpeg"):
            FileEncoder("nope.mp4", 16, 16)


def test_vote_matches_counter_semantics():
    from offmark.dist.vote import shard_range, vote, vote_segments
    rng = np.random.default_rng(3)
    for _ in range(300):
        n, L = int(rng.integers(1, 14)), int(rng.integers(1, 5))
        pats = rng.integers(0, 2, size=(n, L))
        a, b = vote(pats), orc.vote(list(pats))
        assert np.array_equal(a[0], b[0]) and abs(a[1] - b[1]) < 1e-12
    assert vote(np.zeros((0, 8))) == (None, None)
    segs = vote_segments(np.array([[0, 1], [0, 1], [1, 1], [1, 0]]), np.array([0, 0, 0, 1]))
    assert segs[0][0].tolist() == [0, 1] and abs(segs[0][1] - 2 / 3) < 1e-12 and segs[1][0].tolist() == [1, 0]
    covered = []
    for rank in range(5):
        a, b = shard_range(23, rank, 5)
        covered += list(range(a, b))
    assert covered == list(range(23))


def test_payloads_from_counts_torch_matches_degenerate_counts():
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.vote import payloads_from_counts
    rng = np.random.default_rng(5)
    for (L, N) in [(8, 32400), (5, 37), (13, 300)]:
        deg = DeShuffler(key=7).set_shape((L,))
        lens = np.array([(N - i + L - 1) // L for i in range(L)])
        counts = (rng.random((20, L)) * lens).astype(np.int32)
        ref = deg.degenerate_counts(counts, N)
        got = payloads_from_counts(torch.from_numpy(counts), N, deg.payload_idx).numpy()
        assert np.array_equal(got, ref)


def test_fingerprint_bookkeeping():
    from offmark import fingerprint as fp
    assert fp.payload_for_segment(5).tolist() == [0, 0, 0, 0, 0, 1, 0, 1]
    assert fp.payload_for_segment(261).tolist() == fp.payload_for_segment(5).tolist()      # wraps at 256
    assert fp.payload_for_segment(3, 2).tolist() == [0, 0, 1, 1, 0, 0, 1, 0]
    assert fp.payload_for_segment(19, 18).tolist() == fp.payload_for_segment(3, 2).tolist()  # both wrap at 16
    assert fp.decode_pattern(np.array([0, 0, 1, 1, 0, 0, 1, 0])) == (3, 2)
    assert fp.decode_pattern([1, 0, 1]) == (None, None) and fp.decode_pattern(None) == (None, None)
    assert fp.select_copies("01201201", 8, 3) == [0, 1, 2, 0, 1, 2, 0, 1]
    assert fp.select_copies("57", 2, 3) == [2, 1]
    with pytest.raises(ValueError):
        fp.select_copies("01", 3, 3)
    assert fp.view_to_copies(5, 3, 4) == [0, 0, 1, 2] and fp.view_to_copies(0, 3, 2) == [0, 0]
    votes = {0: (fp.payload_for_segment(0, 1), 1.0), 1: (fp.payload_for_segment(1, 2), 0.9),
             2: (fp.payload_for_segment(7, 0), 0.8)}                                         # segment 2 decodes as 7: reject
    assert fp.identify_copies(votes) == [1, 2, None]


def build_abi_demo(out_path):
    """examples/abi_demo.c with the plain C compiler against the in-tree library and the HIP runtime."""
    import __graft_entry__ as ge
    libdir = os.path.dirname(ge.LIB)
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-O2", os.path.join(ROOT, "examples", "abi_demo.c"), "-I", os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L", libdir, "-L/opt/rocm/lib", "-loffmark_hip", "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", out_path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_header_is_plain_c_and_the_c_host_example_links(tmp_path):
    """include/offmark_hip.h is a C header (no C++, no torch types): a C99 translation unit that only includes it compiles
    with -pedantic, and the stand-alone C host (examples/abi_demo.c) links against the library and the HIP runtime."""
    tu = tmp_path / "only_header.c"
    tu.write_text('#include "offmark_hip.h"\nint main(void) { return ofmk_version() == OFMK_ABI_VERSION ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(tu)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    build_abi_demo(str(tmp_path / "abi_demo"))


def test_pipeline_host_side_read_ahead_and_writer_protocol():
    """Host logic of offmark/video/pipeline.py that needs no GPU: the read-ahead thread over every kind of reader
    (batches, read_batch_into, bare read()), order and ragged tail, error hand-over, size check, threaded host copy,
    the frame-shape rules of the pix_fmt route, PeekedReader, and ArrayFrameWriter's plain (copying) form."""
    import queue
    from offmark.video import pipeline as pl
    from offmark.video.frame_reader import ArrayFrameReader
    from offmark.video.frame_writer import ArrayFrameWriter
    frames = np.arange(11 * 4 * 6 * 3, dtype=np.uint8).reshape(11, 4, 6, 3)

    class Bare:
        def __init__(self, fr, fail_at=None):
            self.fr, self.i, self.fail_at = list(fr), 0, fail_at

        def read(self):
            if self.i == self.fail_at:
                raise IOError("boom")
            self.i += 1
            return self.fr[self.i - 1] if self.i <= len(self.fr) else None

        def close(self):
            pass

    class Into(Bare):
        def read_batch_into(self, buf):
            n = 0
            while n < len(buf):
                f = self.read()
                if f is None:
                    break
                buf[n] = f
                n += 1
            return n

    def drain(reader, batch=4, staging=3):
        ra = pl._ReadAhead(reader, batch, (4, 6, 3), np.uint8, [np.empty((batch, 4, 6, 3), np.uint8) for _ in range(staging)])
        ra.start()
        got = []
        while True:
            item = ra.ready.get(timeout=30)
            if item is None or isinstance(item, BaseException):
                ra.shutdown()
                return got, item
            view, st = item
            got.append(view.copy())
            if st is not None:
                ra.free.put(st)

    for reader in (ArrayFrameReader(frames), Bare(frames), Into(frames)):
        got, end = drain(reader)
        assert end is None and [len(g) for g in got] == [4, 4, 3] and np.array_equal(np.concatenate(got), frames)
    got, end = drain(Bare(frames, fail_at=6))
    # every frame read before the failure comes through first, the partial batch included (the reference's loop
    # would have processed them all), then the exception
    assert isinstance(end, IOError) and [len(g) for g in got] == [4, 2] and np.array_equal(np.concatenate(got), frames[:6])
    got, end = drain(Bare(list(frames[:5]) + [np.zeros((5, 6, 3), np.uint8)]))
    assert isinstance(end, ValueError) and "expected" in str(end)
    got, end = drain(Bare([]))
    assert got == [] and end is None
    big = np.random.default_rng(0).integers(0, 256, (40, 512, 512, 3), dtype=np.uint8)        # 31 MB: the threaded path
    dst = np.empty_like(big)
    pl.host_copy(dst, big)
    assert np.array_equal(dst, big)
    assert pl.frame_shape("rgb24", 240, 320) == (240, 320, 3) and pl.frame_shape("yuv420p", 240, 320) == (360, 320)
    with pytest.raises(ValueError):
        pl.frame_shape("nv12", 241, 320)
    with pytest.raises(ValueError):
        pl.pix_fmt_of(type("R", (), {"pix_fmt": "yuv444p"})())
    assert pl.pix_fmt_of(object()) == "rgb24"
    pk = pl.PeekedReader(Bare(frames[:3]))
    assert (pk.height, pk.width) == (4, 6) and np.array_equal(np.stack([pk.read(), pk.read(), pk.read()]), frames[:3]) and pk.read() is None
    w = ArrayFrameWriter()
    assert w.reserve(4) is None                                  # no page-locked block: the pipeline falls back to write_batch
    buf = frames[:4].copy()
    w.write_batch(buf)
    buf[:] = 0                                                   # the caller reuses its buffer: the writer kept a copy
    assert np.array_equal(np.stack(w.frames), frames[:4])
    assert np.array_equal(w.array(), frames[:4])                  # array() without a block: the stacked stream, not an error
    w.write_batch(frames[4:6].astype(np.int32))                  # a non-uint8 batch is cast, as the reference's writer casts
    assert np.array_equal(w.array(), frames[:6]) and w.array().dtype == np.uint8
    assert ArrayFrameWriter().array().shape[0] == 0
    assert queue.Queue                                           # (imported for the timeout above)


def test_file_decoder_and_encoder_plumbing_with_a_stand_in_ffmpeg(fake_ffmpeg, tmp_path):
    """FileDecoder / FileEncoder (reference frame_reader.py:28-69, frame_writer.py:23-50) drive `ffmpeg` child processes.
    No ffmpeg exists on the test boxes, so a test double on PATH (tests/conftest.py: raw frames behind a one-line header,
    formats passed through) stands in for the binaries: what is tested is this package's plumbing -- the probe, the command
    lines, frame sizes for rgb24 and yuv420p, read(), read_batch(), read_batch_into() straight from the pipe, end of
    stream, write() / write_batch() and close()."""
    from offmark.video.frame_reader import FileDecoder
    from offmark.video.frame_writer import FileEncoder
    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, (7, 16, 24, 3), dtype=np.uint8)
    path = fake_ffmpeg(tmp_path / "in.raw", frames)
    r = FileDecoder(path)
    assert (r.width, r.height, r.pix_fmt) == (24, 16, "rgb24")
    assert np.array_equal(r.read(), frames[0])
    assert np.array_equal(r.read_batch(2), frames[1:3])
    buf = np.empty((3, 16, 24, 3), np.uint8)
    assert r.read_batch_into(buf) == 3 and np.array_equal(buf, frames[3:6])
    assert r.read_batch_into(buf) == 1 and np.array_equal(buf[0], frames[6])       # ragged tail
    assert r.read_batch_into(buf) == 0 and r.read() is None
    r.close()
    planes = rng.integers(0, 256, (4, 24, 24), dtype=np.uint8)                     # 16x24 yuv420p frames: [H*3/2, W]
    rp = FileDecoder(fake_ffmpeg(tmp_path / "p.raw", planes, "yuv420p"), pix_fmt="yuv420p")
    assert (rp.width, rp.height) == (24, 16) and np.array_equal(rp.read_batch(9), planes)
    rp.close()
    out = str(tmp_path / "out.raw")
    w = FileEncoder(out, 24, 16)
    w.write(frames[0])
    w.write_batch(frames[1:4])
    w.write(frames[4][:, ::-1][:, ::-1])                                           # a non-contiguous view is written whole too
    w.close()
    back = FileDecoder(out)
    assert np.array_equal(back.read_batch(99), frames[:5])
    back.close()


def test_pipeline_batches_are_bounded_in_bytes():
    from offmark.video.pipeline import batch_size, frame_shape
    assert batch_size(64, frame_shape("rgb24", 1080, 1920)) == 64            # 398 MB: the caller's choice stands
    assert batch_size(64, frame_shape("rgb24", 2160, 3840)) == 21            # 4K: 64 frames would be 1.6 GB per buffer, 9 buffers
    assert batch_size(64, frame_shape("yuv420p", 2160, 3840)) == 43
    assert batch_size(0, (8, 8, 3)) == 1 and batch_size(5, (8, 8, 3)) == 5


def test_balanced_chunks():
    """engine.balanced_chunk: fewest chunks under the cap, all nearly equal (384 under 345 -> 192 + 192)."""
    import importlib
    E = importlib.import_module("offmark.engine")
    assert E.balanced_chunk(384, 345) == 192
    assert E.balanced_chunk(300, 345) == 300 and E.balanced_chunk(1000, 86) == 84 and E.balanced_chunk(100, 40) == 34
    assert E.balanced_chunk(1, 5) == 1 and E.balanced_chunk(7, 1) == 1 and E.balanced_chunk(0, 4) == 1
    for n in range(1, 400, 7):
        for cap in (1, 3, 17, 64, 345):
            c = E.balanced_chunk(n, cap)
            k = -(-n // c)
            assert 1 <= c <= max(cap, 1) and k == -(-n // min(cap, n)) and n - (k - 1) * c > c - k      # same chunk count as the cap gives; last chunk within k of the rest


def test_chunk_choice_never_exceeds_what_the_library_launches():
    """with the 8 GiB byte cap tiny frames gave a Python-side chunk above the library's 65 535 frames per launch (gridDim.y,
    csrc kMaxChunk), so the workspace, the timing pool and the order bucket were computed for a launch shape that never runs.
    The engine's chunk helpers and the static tile-order rule restated for reporting (include/offmark_hip.h)."""
    import importlib
    E = importlib.import_module("offmark.engine")
    from offmark import _hip
    assert E.default_chunk_frames(8, 8) == 65535 and E.default_chunk_frames(16, 16) == 65535
    assert E.balanced_chunk(200000, 10 ** 9) == 50000 and E.balanced_chunk(65535, 10 ** 9) == 65535      # 4 equal chunks under the clamp
    assert E.default_chunk_frames(1080, 1920) == (8 << 30) // (1080 * 1920 * 3)
    assert _hip.XCD_TILES_MIN_BYTES == 192 * 1080 * 1920 * 3
    header = open(os.path.join(ROOT, "include", "offmark_hip.h")).read()
    assert f"#define OFMK_XCD_TILES_MIN_BYTES {_hip.XCD_TILES_MIN_BYTES}ull" in header
    assert E.static_tile_order(_hip.XCD_TILES_MIN_BYTES) == "xcd" and E.static_tile_order(_hip.XCD_TILES_MIN_BYTES - 1) == "linear"
    assert (_hip.F_SEPARATE_DETECT, _hip.F_LINEAR_TILES, _hip.F_XCD_TILES, _hip.F_PARTIAL_COUNTS) == (1, 2, 4, 8)
    for name, value in (("OFMK_F_SEPARATE_DETECT", 1), ("OFMK_F_LINEAR_TILES", 2), ("OFMK_F_XCD_TILES", 4), ("OFMK_F_PARTIAL_COUNTS", 8)):
        assert f"#define {name} {value}u" in header
