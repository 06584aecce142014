"""The C restatement (oracle/offmark_oracle.c) must agree BIT FOR BIT with the NumPy oracle and with the
vectors captured by running the reference's own modules over the restated cv2 primitives (they pin the reference's
logic; OpenCV's float rounding is parity-unpinned, see tests/test_oracle_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import c_oracle
import offmark_oracle as orc
from conftest import GOLDEN, golden_cases

P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])


@pytest.mark.parametrize("case", golden_cases())
def test_c_oracle_reproduces_reference_logic_vectors(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    alpha = float(g["alpha"])
    marked, _ = c_oracle.mark_frames(g["frame"][None], g["wm"], alpha=alpha, legacy=False)    # goldens: numpy 2 semantics
    assert np.array_equal(marked[0], g["marked"])
    bits, _ = c_oracle.check_frames(g["marked"][None], alpha=alpha, legacy=False)
    assert np.array_equal(bits[0], g["raw_bits"].reshape(-1))


@pytest.mark.parametrize("legacy", [True, False])
@pytest.mark.parametrize("shape,seed", [((240, 320), 1001), ((1080, 1920), 2000), ((30, 44), 5), ((2160, 3840), 3001)])
def test_c_oracle_equals_numpy_oracle(shape, seed, legacy):
    if shape[0] > 1080 and not legacy:
        pytest.skip("one 4K comparison is enough")
    H, W = shape
    frame = orc.synthetic_frame(H, W, seed)
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
    promo = "legacy" if legacy else "nep50"
    enc = orc.DctEncoderOracle(alpha=20, promotion=promo)
    enc.read_wm(wm)
    ref = orc.mark_frame(frame, enc)
    got, _ = c_oracle.mark_frames(frame[None], wm, alpha=20, legacy=legacy)
    assert np.array_equal(got[0], ref)
    ref_bits = orc.check_frame(ref, orc.DctDecoderOracle(alpha=20, promotion=promo))
    bits, _ = c_oracle.check_frames(ref[None], alpha=20, legacy=legacy)
    assert np.array_equal(bits[0], ref_bits.reshape(-1))


def test_c_oracle_threads_give_identical_results():
    frames = np.stack([orc.synthetic_frame(240, 320, 1001 + i) for i in range(6)])
    wm = orc.shuffle_generate(P8, (1, 1200), 0)
    a, _ = c_oracle.mark_frames(frames, wm, threads=1)
    b, used = c_oracle.mark_frames(frames, wm, threads=4)
    assert used == 4 and np.array_equal(a, b)


def test_full_natural_frame_digests_match_the_reference_run():
    """The reference's own 1920x1080 JPEG frame: NumPy oracle, C oracle and the reference modules' run
    (stored as SHA-256 digests by tools/make_golden.py) must all produce the same marked frame and raw bits."""
    import hashlib
    from conftest import natural_frame
    g = np.load(os.path.join(GOLDEN, "frame63_full_digest.npz"))
    nat = natural_frame()
    assert nat.shape == (1080, 1920, 3)
    wm = orc.shuffle_generate(P8, (1, 32400), 0)
    enc = orc.DctEncoderOracle(alpha=20, promotion="nep50")
    enc.read_wm(wm)
    marked = orc.mark_frame(nat, enc)
    assert hashlib.sha256(marked.tobytes()).digest() == g["marked_sha256"].tobytes()
    raw = orc.check_frame(marked, orc.DctDecoderOracle(alpha=20, promotion="nep50"))
    assert hashlib.sha256(raw.astype(np.uint8).tobytes()).digest() == g["raw_bits_sha256"].tobytes()
    assert abs(np.mean(raw.reshape(-1) != wm.reshape(-1)) - float(g["raw_ber"])) < 1e-12
    assert np.array_equal(orc.deshuffle(raw, 8, 0), g["degenerated"]) and np.array_equal(g["degenerated"], P8)
    cm, _ = c_oracle.mark_frames(nat[None], wm, alpha=20, legacy=False)
    cb, _ = c_oracle.check_frames(cm, alpha=20, legacy=False)
    assert np.array_equal(cm[0], marked) and np.array_equal(cb[0], raw.reshape(-1))
