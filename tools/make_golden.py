"""Generate tests/golden/*.npz by running the reference's own, unmodified modules.

Run in the build container only (needs /root/reference):  python tools/make_golden.py

* numpy-only reference modules (shuffler, grayscale, de_shuffler, de_grayscale) are imported
  and run as they are -> fully pinned vectors.
* dct_encoder / dct_decoder / video.embedder import ``cv2``; tools/standins supplies
  dct / idct / cvtColor from the oracle primitives (OpenCV arithmetic itself unpinned).
* ``--standin scipy`` runs the 16 DCT cases and the 13 blk-4 DwtDctSvd cases a second time with
  tools/standins_scipy -- ``scipy.fft.dctn/idctn(norm="ortho")``, plain-NumPy ``cvtColor``, closed-form
  Haar: primitives the oracle did NOT supply -- into tests/golden_scipy/.  Those vectors differ from the
  oracle's by float rounding only; tests/test_golden_scipy.py compares at the stated budgets.
The reference's text is never copied: only inputs and outputs are stored.
The container's numpy is 2.x, so the captured texture-mask values follow NEP 50 promotion
(``promotion="nep50"`` in the oracle).
"""
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STANDIN = "oracle"
if "--standin" in sys.argv:
    STANDIN = sys.argv[sys.argv.index("--standin") + 1]
    del sys.argv[sys.argv.index("--standin"):sys.argv.index("--standin") + 2]
    assert STANDIN in ("oracle", "scipy"), STANDIN
sys.path.insert(0, os.path.join(HERE, "standins_scipy" if STANDIN == "scipy" else "standins"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference/src")

# video/embedder.py imports only cv2/numpy/logging; frame_reader/writer need ffmpeg -> not imported.
import cv2  # noqa: E402  (the stand-in)
import offmark_oracle as orc  # noqa: E402
from offmark.degenerator.de_grayscale import DeGrayScale  # noqa: E402
from offmark.degenerator.de_shuffler import DeShuffler  # noqa: E402
from offmark.embed.dct_encoder import DctEncoder  # noqa: E402
from offmark.extract.dct_decoder import DctDecoder  # noqa: E402
from offmark.generator.grayscale import GrayScale  # noqa: E402
from offmark.generator.shuffler import Shuffler  # noqa: E402
from offmark.video.embedder import Embedder  # noqa: E402
from offmark.embed.dwt_dct_svd_encoder import DwtDctSvdEncoder  # noqa: E402  (needs the pywt + cv2 stand-ins)
from offmark.extract.dwt_dct_svd_decoder import DwtDctSvdDecoder  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden_scipy" if STANDIN == "scipy" else "golden")
os.makedirs(OUT, exist_ok=True)
logging.disable(logging.CRITICAL)


def run_case(name, frame, payload, key, alpha, image_payload=False, store_yuv=False):
    h, w, _ = frame.shape
    enc = DctEncoder(alpha=alpha)
    dec = DctDecoder(alpha=alpha)
    cap = enc.wm_capacity((h, w, 3))
    if image_payload:
        gen, deg = GrayScale(key=key), DeGrayScale(key=key)
    else:
        gen, deg = Shuffler(key=key), DeShuffler(key=key)
    wm = gen.generate_wm(payload, cap)
    enc.read_wm(wm)
    deg.set_shape(payload.shape)

    yuv_in = cv2.cvtColor(frame.astype(np.float32), cv2.COLOR_BGR2YUV)
    lum = enc.luminance_mask(yuv_in[:, :, 0])
    tex = enc.texture_mask(yuv_in[:, :, 0])
    yuv_out = enc.encode(yuv_in.copy())
    marked = Embedder(None, enc, None)._Embedder__mark_frame(frame)       # embedder.py:33-39
    # extractor.py:30-34 (its private method only logs; same three calls)
    yuv_rx = cv2.cvtColor(marked.astype(np.float32), cv2.COLOR_BGR2YUV)
    raw_bits = dec.decode(yuv_rx)
    out = deg.degenerate(raw_bits)
    raw_bits_clean = dec.decode(yuv_out.copy())        # decode straight from the f32 encoder output
    d = dict(frame=frame, payload=np.asarray(payload), key=np.int64(key), alpha=np.float64(alpha),
             image_payload=np.bool_(image_payload), wm=wm, lum_mask=lum, tex_mask=tex,
             marked=marked, raw_bits=raw_bits, raw_bits_clean=raw_bits_clean, degenerated=out,
             perm=deg.payload_idx)
    if store_yuv:
        d.update(yuv_in=yuv_in, yuv_out=yuv_out)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    ok = np.array_equal(np.asarray(out).reshape(-1) // (255 if image_payload else 1),
                        (np.asarray(payload).reshape(-1) > 127).astype(np.uint8) if image_payload
                        else np.asarray(payload).reshape(-1))
    print(f"{name:28s} {h}x{w} L={np.asarray(payload).size:4d} key={key} alpha={alpha} "
          f"raw_ber={np.mean(raw_bits.reshape(-1)[:wm.size] != wm.reshape(-1)):.4f} payload_ok={ok}")


def run_svd_case(name, frame, payload, key, store_yuv=False, scales=None, blk=None):
    """mark.py / detect.py's codec pair (DwtDctSvdEncoder / DwtDctSvdDecoder) on one frame.
    scales: per-channel quantisation steps (dwt_dct_svd_encoder.py:6,19-26); None = the reference's default [0,15,0].
    blk: LL block size (dwt_dct_svd_encoder.py:6,29-40); None = the reference's default 4."""
    h, w, _ = frame.shape
    kw = {}
    if scales is not None:
        kw["scales"] = list(scales)
    if blk is not None:
        kw["blk"] = int(blk)
    enc, dec = DwtDctSvdEncoder(**kw), DwtDctSvdDecoder(**kw)
    wm = Shuffler(key=key).generate_wm(payload, enc.wm_capacity((h, w, 3)))
    enc.read_wm(wm)
    deg = DeShuffler(key=key).set_shape(payload.shape)
    yuv_in = cv2.cvtColor(frame.astype(np.float32), cv2.COLOR_BGR2YUV)
    yuv_out = enc.encode(yuv_in.copy())
    marked = Embedder(None, enc, None)._Embedder__mark_frame(frame)
    raw_bits = dec.decode(cv2.cvtColor(marked.astype(np.float32), cv2.COLOR_BGR2YUV))
    d = dict(frame=frame, payload=np.asarray(payload), key=np.int64(key), wm=wm, marked=marked, raw_bits=raw_bits,
             raw_bits_clean=dec.decode(yuv_out.copy()), degenerated=deg.degenerate(raw_bits))
    if scales is not None:
        d["scales"] = np.asarray(scales, dtype=np.float64)
    if blk is not None:
        d["blk"] = np.int64(blk)
    if store_yuv:
        d.update(yuv_in=yuv_in, yuv_out=yuv_out)
    np.savez_compressed(os.path.join(OUT, "svd_" + name + ".npz"), **d)
    nb = min(raw_bits.size, wm.size)                       # blk=8: the decoder returns a quarter as many bits
    print(f"svd_{name:24s} {h}x{w} raw_ber={np.mean(raw_bits.reshape(-1)[:nb] != wm.reshape(-1)[:nb]):.4f} "
          f"payload_ok={np.array_equal(d['degenerated'], payload)}")


def svd_scale_cases():
    """Round 2: per-channel scales of the DwtDctSvd codec (any subset of the three channels marked)."""
    from PIL import Image
    P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
    nat = np.asarray(Image.open("/root/reference/tests/media/imgs/frame63.jpeg").convert("RGB"))
    run_svd_case("scales_10_15_20_syn_64x96", orc.synthetic_frame(64, 96, 2), P8, 0, store_yuv=True, scales=(10, 15, 20))
    run_svd_case("scales_12_0_0_syn_64x96", orc.synthetic_frame(64, 96, 3), P8, 0, store_yuv=True, scales=(12, 0, 0))
    run_svd_case("scales_0_9_25_syn_36x52", orc.synthetic_frame(36, 52, 6), P8, 7, store_yuv=True, scales=(0, 9, 25))
    run_svd_case("scales_0_0_30_syn_240x320", orc.synthetic_frame(240, 320, 1001), P8, 0, scales=(0, 0, 30))
    run_svd_case("scales_8_22_8_frame63_crop0", np.ascontiguousarray(nat[300:428, 600:728]), P8, 0, scales=(8, 22, 8))


def svd_blk8_cases():
    """Round 3: DwtDctSvd with blk=8 (16x16 pixel tiles; the encoder consumes the first quarter of the watermark, the decoder
    returns row*col//256 bits: dwt_dct_svd_encoder.py:29-40, dwt_dct_svd_decoder.py:14)."""
    from PIL import Image
    P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
    nat = np.asarray(Image.open("/root/reference/tests/media/imgs/frame63.jpeg").convert("RGB"))
    run_svd_case("blk8_syn_64x96", orc.synthetic_frame(64, 96, 2), P8, 0, store_yuv=True, blk=8)
    run_svd_case("blk8_syn_240x320", orc.synthetic_frame(240, 320, 1001), P8, 0, blk=8)
    run_svd_case("blk8_syn_36x52", orc.synthetic_frame(36, 52, 6), P8, 7, store_yuv=True, blk=8)          # LL 18x26: 2x3 tiles, fringe untouched
    run_svd_case("blk8_frame63_crop0", np.ascontiguousarray(nat[300:428, 600:728]), P8, 0, blk=8)
    run_svd_case("blk8_scales_10_15_20_syn_64x96", orc.synthetic_frame(64, 96, 3), P8, 0, store_yuv=True, scales=(10, 15, 20), blk=8)
    run_svd_case("blk8_edge_black_64x64", np.zeros((64, 64, 3), np.uint8), P8, 0, blk=8)
    run_svd_case("blk8_edge_white_64x64", np.full((64, 64, 3), 255, np.uint8), P8, 0, blk=8)


def grayscale_at_scale_digests():
    """Round 2: GrayScale / DeGrayScale with the reference's own 480x270 payload image
    (tests/media/wms/numbers.jpeg, L = 129 600 bits) on full frames.  The frames are synthetic and regenerable, the
    outputs are too large to store: SHA-256 digests of every stage plus the small decoded image.
    1080p: capacity 32 400 < L, so GrayScale warns and truncates and DeGrayScale's means of empty slices are nan
    (de_grayscale.py:17-21): the reference decodes an all-zero image.  4K: capacity == L, one block per payload bit."""
    import hashlib
    import warnings
    from PIL import Image
    img = np.asarray(Image.open("/root/reference/tests/media/wms/numbers.jpeg").convert("L"))
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)      # noqa: E731
    out = dict(payload_shape=np.asarray(img.shape), key=np.int64(3), alpha=np.float64(20))
    for tag, (h, w, seed) in (("1080p", (1080, 1920, 2000)), ("4k", (2160, 3840, 3001))):
        frame = orc.synthetic_frame(h, w, seed)
        enc, dec = DctEncoder(alpha=20), DctDecoder(alpha=20)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            wm = GrayScale(key=3).generate_wm(img, enc.wm_capacity(frame.shape))
        enc.read_wm(wm)
        marked = Embedder(None, enc, None)._Embedder__mark_frame(frame)
        raw = dec.decode(cv2.cvtColor(marked.astype(np.float32), cv2.COLOR_BGR2YUV))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            deg = DeGrayScale(key=3).set_shape(img.shape).degenerate(raw)
        out.update({f"{tag}_seed": np.int64(seed), f"{tag}_warned": np.bool_(len(caught) > 0), f"{tag}_wm_sha256": sha(wm.astype(np.uint8)),
                    f"{tag}_marked_sha256": sha(marked), f"{tag}_raw_bits_sha256": sha(raw.astype(np.uint8)),
                    f"{tag}_degenerated_packed": np.packbits(deg.reshape(-1) > 0), f"{tag}_raw_ber": np.float64(np.mean(raw.reshape(-1) != wm.reshape(-1))),
                    f"{tag}_image_agreement": np.float64(np.mean((deg > 0) == (img > 127)))})
        print(f"grayscale_numbers {tag}: warned={len(caught) > 0} raw_ber={out[f'{tag}_raw_ber']:.4f} "
              f"decoded image agrees with the payload on {out[f'{tag}_image_agreement']:.4f} of its pixels")
    np.savez_compressed(os.path.join(OUT, "grayscale_numbers_digest.npz"), **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--round2":       # add the round-2 fixtures without touching round 1's
        svd_scale_cases()
        grayscale_at_scale_digests()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--round3":
        svd_blk8_cases()
        return
    if STANDIN == "scipy":      # second fixture set: the DCT cases and the blk-4 DwtDctSvd cases only, no digests
        dct_and_svd4_cases()
        svd_scale_cases()
        return
    dct_and_svd4_cases()
    full_frame_digest_and_payload_codecs()
    svd_scale_cases()
    grayscale_at_scale_digests()
    svd_blk8_cases()


def dct_and_svd4_cases():
    P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
    run_svd_case("syn_64x96", orc.synthetic_frame(64, 96, 2), P8, 0, store_yuv=True)
    run_svd_case("syn_240x320", orc.synthetic_frame(240, 320, 1001), P8, 0)
    run_svd_case("syn_30x44", orc.synthetic_frame(30, 44, 5), P8, 0, store_yuv=True)
    run_svd_case("syn_36x52", orc.synthetic_frame(36, 52, 6), P8, 7, store_yuv=True)
    run_svd_case("edge_black_64x64", np.zeros((64, 64, 3), np.uint8), P8, 0)
    run_svd_case("edge_white_64x64", np.full((64, 64, 3), 255, np.uint8), P8, 0)
    P5 = np.array([1, 0, 0, 1, 1])
    run_case("syn_16x24_L8_k0_a20", orc.synthetic_frame(16, 24, 1), P8, 0, 20, store_yuv=True)
    run_case("syn_64x96_L8_k0_a20", orc.synthetic_frame(64, 96, 2), P8, 0, 20, store_yuv=True)
    run_case("syn_64x96_L5_k7_a10", orc.synthetic_frame(64, 96, 3), P5, 7, 10)
    run_case("syn_240x320_L8_k0_a20", orc.synthetic_frame(240, 320, 1001), P8, 0, 20)
    run_case("syn_30x44_L8_k0_a20", orc.synthetic_frame(30, 44, 5), P8, 0, 20, store_yuv=True)
    big = orc.synthetic_frame(1080, 1920, 2000)
    for n, (y, x) in enumerate([(0, 0), (128, 256), (512, 1024), (952, 1792)]):
        run_case(f"syn1080_crop{n}_L8_k0_a20", np.ascontiguousarray(big[y:y + 128, x:x + 128]), P8, 0, 20)
    # natural image shipped with the reference's tests (data file, decoded with Pillow)
    from PIL import Image
    nat = np.asarray(Image.open("/root/reference/tests/media/imgs/frame63.jpeg").convert("RGB"))
    for n, (y, x) in enumerate([(300, 600), (700, 1200)]):
        run_case(f"frame63_crop{n}_L8_k0_a20", np.ascontiguousarray(nat[y:y + 128, x:x + 128]), P8, 0, 20)
    for n, (y, x) in enumerate([(300, 600), (700, 1200)]):
        run_svd_case(f"frame63_crop{n}", np.ascontiguousarray(nat[y:y + 128, x:x + 128]), P8, 0)
    qr = np.asarray(Image.open("/root/reference/tests/media/wms/qr.jpeg").convert("L"))
    run_case("frame63_crop_qr_k0_a20", np.ascontiguousarray(nat[256:256 + 256, 512:512 + 384]), qr, 0, 20,
             image_payload=True)
    # edge cases
    run_case("edge_black_64x64", np.zeros((64, 64, 3), np.uint8), P8, 0, 20)
    run_case("edge_white_64x64", np.full((64, 64, 3), 255, np.uint8), P8, 0, 20)
    run_case("edge_gray_64x64", np.full((64, 64, 3), 128, np.uint8), P8, 0, 20)
    run_case("edge_const_payload_64x96", orc.synthetic_frame(64, 96, 9), np.ones(8, dtype=np.int64), 0, 20)


def full_frame_digest_and_payload_codecs():
    from PIL import Image
    P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
    nat = np.asarray(Image.open("/root/reference/tests/media/imgs/frame63.jpeg").convert("RGB"))
    # the whole 1920x1080 natural frame: too large to store its outputs, so store their SHA-256 digests
    import hashlib
    enc, dec = DctEncoder(alpha=20), DctDecoder(alpha=20)
    wm = Shuffler(key=0).generate_wm(P8, enc.wm_capacity(nat.shape))
    enc.read_wm(wm)
    marked = Embedder(None, enc, None)._Embedder__mark_frame(nat)
    raw = dec.decode(cv2.cvtColor(marked.astype(np.float32), cv2.COLOR_BGR2YUV))
    out = DeShuffler(key=0).set_shape(P8.shape).degenerate(raw)
    np.savez_compressed(os.path.join(OUT, "frame63_full_digest.npz"), payload=P8, key=np.int64(0), alpha=np.float64(20),
                        marked_sha256=np.frombuffer(hashlib.sha256(marked.tobytes()).digest(), np.uint8),
                        raw_bits_sha256=np.frombuffer(hashlib.sha256(raw.astype(np.uint8).tobytes()).digest(), np.uint8),
                        raw_ber=np.float64(np.mean(raw.reshape(-1) != wm.reshape(-1))), degenerated=out)
    print(f"frame63_full_digest          1080x1920 raw_ber={np.mean(raw.reshape(-1) != wm.reshape(-1)):.4f} payload_ok={np.array_equal(out, P8)}")
    # payload codecs alone (numpy-only reference modules, no stand-in involved)
    rows = {}
    for key in (0, 7, None):
        for L in (5, 8, 13):
            rs = np.random.RandomState(100 + L)
            p = rs.randint(0, 2, size=L)
            for cap in ((1, 300), (1, 32400), (1, 37)):
                tag = f"k{key}_L{L}_c{cap[1]}"
                if key is None:
                    continue  # unseeded -> not reproducible
                wm = Shuffler(key=key).generate_wm(p, cap)
                noisy = wm.astype(np.float64).copy()
                flip = rs.rand(noisy.size) < 0.2
                noisy.reshape(-1)[flip] = 1 - noisy.reshape(-1)[flip]
                back = DeShuffler(key=key).set_shape(p.shape).degenerate(noisy)
                rows[tag + "_payload"] = p
                rows[tag + "_wm"] = wm
                rows[tag + "_noisy"] = noisy
                rows[tag + "_back"] = back
    np.savez_compressed(os.path.join(OUT, "payload_codecs.npz"), **rows)
    print("payload_codecs", len(rows) // 4, "cases")


if __name__ == "__main__":
    main()
