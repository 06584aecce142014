"""GPU parity for the planar 8-bit YUV 4:2:0 entry points (I420 and NV12 in, the same out).

The reference has ffmpeg convert between its rgb24 pipes and yuv420p files (src/offmark/video/frame_reader.py:42-64,
frame_writer.py:33-34).  swscale is not available, so the conversion is BUILD-DEFINED (csrc/planar_kernels.hiph,
restated in oracle/offmark_oracle.py: yuv420_to_rgb / rgb_to_yuv420) and what is pinned here is:
  * the conversion kernels against the oracle's restatement .................. bit-exact
  * fused planar kernels against the unfused chain convert -> RGB engine -> convert ... bit-exact
  * the whole planar path against the oracle pipeline (oracle conversion around the oracle's mark_frame /
    check_frame) ....... the RGB path's budgets (tests/test_gpu_parity.py), applied to the written planes
"""
import numpy as np
import pytest

import offmark_oracle as orc

pytestmark = pytest.mark.gpu
P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
LAYOUTS = ["i420", "nv12"]


@pytest.fixture(scope="module")
def eng():
    import torch
    from offmark.engine import DctEngine
    torch.cuda.set_device(0)
    return DctEngine()


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def planes_of(rgb_frames, layout):
    """Host: oracle conversion of u8 RGB frames [n,H,W,3] to the flat plane layout [n, 1.5*H*W]."""
    return np.stack([orc.pack_yuv420(*orc.rgb_to_yuv420(f), layout) for f in rgb_frames])


@pytest.mark.parametrize("layout", LAYOUTS)
def test_conversion_kernels_equal_the_oracle_restatement(eng, layout):
    rng = np.random.default_rng(5)
    for (H, W, n) in [(16, 24, 3), (64, 96, 2), (240, 320, 2)]:
        # every byte value in every plane, including out-of-gamut combinations that exercise the clip
        raw = rng.integers(0, 256, (n, H * W * 3 // 2), dtype=np.uint8)
        raw[0, : H * W: 7] = 255
        raw[0, 1: H * W: 7] = 0
        got = eng.yuv420_to_rgb(cuda(raw), H, W, layout).cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], orc.yuv420_to_rgb(*orc.unpack_yuv420(raw[i], H, W, layout))), (H, W, i)
        rgb = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
        rgb[0, ::3] = 255
        rgb[0, 1::3] = 0
        back = eng.rgb_to_yuv420(cuda(rgb), layout).cpu().numpy()
        assert np.array_equal(back, planes_of(rgb, layout)), (H, W)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_fused_planar_path_equals_the_unfused_chain_bit_for_bit(eng, layout):
    import torch
    from offmark import _hip
    from offmark.synthetic import synthetic_frames
    for (H, W, n) in [(240, 320, 9), (1080, 1920, 4), (8, 8, 1), (16, 264, 3)]:
        N = H * W // 64
        rgb = synthetic_frames(n, H, W, seed=300 + H)
        planes = eng.rgb_to_yuv420(rgb, layout)
        payloads = np.stack([[int(b) for b in format(s + 1, "08b")] for s in range(3)])
        wm = np.stack([orc.shuffle_generate(p, (N,), 0) for p in payloads])
        rows = (np.arange(n) % 3).astype(np.int32)
        # unfused: planes -> RGB -> RGB engine -> planes; then read the written planes back
        mid = eng.yuv420_to_rgb(planes, H, W, layout)
        ref_out = eng.rgb_to_yuv420(eng.embed(mid, wm, wm_row=rows), layout)
        ref_counts, ref_bits = eng.detect(eng.yuv420_to_rgb(ref_out, H, W, layout), 8, want_bits=True)
        out = eng.embed_yuv420(planes, H, W, wm, wm_row=rows, layout=layout)
        assert torch.equal(out, ref_out), (H, W)
        out2, counts, bits = eng.embed_detect_yuv420(planes, H, W, wm, 8, wm_row=rows, want_bits=True, layout=layout)
        assert torch.equal(out2, ref_out) and torch.equal(counts, ref_counts) and torch.equal(bits, ref_bits), (H, W)
        c3, b3 = eng.detect_yuv420(out, H, W, 8, want_bits=True, layout=layout)
        assert torch.equal(c3, ref_counts) and torch.equal(b3, ref_bits)
        sep = type(eng)(opts=_hip.Opts(_hip.F_SEPARATE_DETECT, 0, None), chunk_frames=2)      # separate detect, several chunks
        out4, c4, b4 = sep.embed_detect_yuv420(planes, H, W, wm, 8, wm_row=rows, want_bits=True, layout=layout)
        assert torch.equal(out4, ref_out) and torch.equal(c4, ref_counts) and torch.equal(b4, ref_bits)
        inplace = planes.clone()
        eng.embed_yuv420(inplace, H, W, wm, wm_row=rows, out=inplace, layout=layout)
        assert torch.equal(inplace, ref_out)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_planar_path_against_the_oracle_pipeline(eng, layout):
    """planes -> (oracle conversion) -> oracle mark_frame -> (oracle conversion) -> planes, and the oracle's detect
    of those planes, against the fused kernels.  Sign-ambiguous blocks (|C21| <= 1e-3 in the oracle) are excluded from
    the sample comparison exactly as in the RGB tests."""
    from offmark.degenerator.de_shuffler import DeShuffler
    for (H, W, seed) in [(240, 320, 1001), (1080, 1920, 2001)]:
        N = H * W // 64
        rgb0 = orc.synthetic_frame(H, W, seed)
        planes = orc.pack_yuv420(*orc.rgb_to_yuv420(rgb0), layout)
        rgb = orc.yuv420_to_rgb(*orc.unpack_yuv420(planes, H, W, layout))           # what the marker sees
        wm = orc.shuffle_generate(P8, (1, N), 0)
        enc = orc.DctEncoderOracle(alpha=20)
        enc.read_wm(wm)
        ref_rgb = orc.mark_frame(rgb, enc)
        ref_planes = orc.pack_yuv420(*orc.rgb_to_yuv420(ref_rgb), layout)
        ref_seen = orc.yuv420_to_rgb(*orc.unpack_yuv420(ref_planes, H, W, layout))   # what a reader of the planes sees
        ref_bits = orc.check_frame(ref_seen, orc.DctDecoderOracle(alpha=20)).reshape(-1)
        out, counts, bits = eng.embed_detect_yuv420(cuda(planes[None]), H, W, wm, 8, want_bits=True, layout=layout)
        got = out[0].cpu().numpy()
        ok = np.abs(enc.debug["c21_pre"]) > 1e-3                                         # sign-determined blocks
        gy, gu, gv = orc.unpack_yuv420(got, H, W, layout)
        ry, ru, rv = orc.unpack_yuv420(ref_planes, H, W, layout)
        my = np.kron(ok, np.ones((8, 8), bool))
        mc = np.kron(ok, np.ones((4, 4), bool))
        d = np.concatenate([np.abs(gy.astype(int) - ry.astype(int))[my], np.abs(gu.astype(int) - ru.astype(int))[mc],
                            np.abs(gv.astype(int) - rv.astype(int))[mc]])
        assert d.max() <= 1 and (d > 0).sum() <= max(1, int(1e-5 * d.size)), (int(d.max()), int((d > 0).sum()), d.size)
        # detect on the oracle's planes (isolates the detect path), and the verify of our own planes on determined blocks
        c2, b2 = eng.detect_yuv420(cuda(ref_planes[None]), H, W, 8, want_bits=True, layout=layout)
        assert (b2[0].cpu().numpy() != ref_bits).sum() <= max(1, int(1e-4 * N))
        own = bits[0].cpu().numpy()
        assert (own[ok.reshape(-1)] != ref_bits[ok.reshape(-1)]).sum() <= max(1, int(1e-4 * N))
        deg = DeShuffler(key=0).set_shape((8,))
        assert np.array_equal(deg.degenerate_counts(c2[0].cpu().numpy(), N), orc.deshuffle(ref_bits, 8, 0))
        # the payload survives the 4:2:0 subsampling of its own chroma channel
        assert np.array_equal(deg.degenerate_counts(counts[0].cpu().numpy(), N), P8)
        print(f"{layout} {W}x{H}: raw bit error rate after 4:2:0 = {(own != wm.reshape(-1)).mean():.3f} "
              f"(oracle {(ref_bits != wm.reshape(-1)).mean():.3f})")


def test_planar_argument_checks(eng):
    import torch
    from offmark import _hip
    lib = _hip.load()
    s = _hip.current_stream()
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    ws = eng.workspace(16, 16, 1)
    wm = torch.zeros(4, dtype=torch.uint8, device="cuda")
    args = (wm.data_ptr(), 1, None, 20.0, 0, ws.data_ptr(), ws.numel(), s, None)
    assert lib.ofmk_embed_yuv420(buf.data_ptr(), buf.data_ptr(), 0, 1, 12, 16, *args) == -1      # H not a multiple of 8
    assert b"multiples of 8" in lib.ofmk_last_error()
    assert lib.ofmk_embed_yuv420(buf.data_ptr(), buf.data_ptr(), 2, 1, 16, 16, *args) == -1      # unknown layout
    assert lib.ofmk_embed_yuv420(buf.data_ptr() + 1, buf.data_ptr(), 0, 1, 16, 16, *args) == -1  # unaligned
    assert lib.ofmk_yuv420_to_rgb8(None, buf.data_ptr(), 0, 1, 16, 16, s, None) == -1
    with pytest.raises(ValueError):
        eng.embed_yuv420(buf[: 16 * 16 * 3 // 2].view(1, -1)[:, :-8], 16, 16, np.zeros((1, 4)))


def test_plugin_classes_on_planes(eng):
    """DctEncoder / DctDecoder convenience entries for planar frames, and the opts validation of the C ABI."""
    import torch
    from offmark import _hip
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.embed.dct_encoder import DctEncoder
    from offmark.extract.dct_decoder import DctDecoder
    from offmark.generator.shuffler import Shuffler
    from offmark.synthetic import synthetic_frames
    H, W, n = 240, 320, 5
    planes = eng.rgb_to_yuv420(synthetic_frames(n, H, W, seed=41), "nv12")
    enc = DctEncoder(alpha=20)
    enc.read_wm(Shuffler(key=0).generate_wm(P8, enc.wm_capacity((H, W, 3))))
    marked = enc.encode_planes_yuv420(planes, H, W, layout="nv12")
    assert torch.equal(marked, eng.embed_yuv420(planes, H, W, enc.wm[None], layout="nv12"))
    counts, _ = DctDecoder(alpha=20).decode_planes_yuv420(marked, H, W, 8, layout="nv12")
    got = DeShuffler(key=0).set_shape((8,)).degenerate_counts(counts.cpu().numpy(), H * W // 64)
    assert np.array_equal(got, np.tile(P8, (n, 1)))
    lib = _hip.load()
    ws = eng.workspace(H, W, n)
    bad = _hip.Opts(64, 0, None)          # an unknown flag bit
    wm = torch.zeros(H * W // 64, dtype=torch.uint8, device="cuda")
    assert lib.ofmk_embed_yuv420(planes.data_ptr(), marked.data_ptr(), 1, n, H, W, wm.data_ptr(), 1, None, 20.0, 0, ws.data_ptr(),
                                 ws.numel(), _hip.current_stream(), bad) == -1
    assert b"ofmk_opts" in lib.ofmk_last_error()
