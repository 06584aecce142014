"""DwtDctSvdDecoder on the MI355X.  Mirrors offmark.extract.dwt_dct_svd_decoder.DwtDctSvdDecoder
(reference src/offmark/extract/dwt_dct_svd_decoder.py:5-37): decode(yuv) -> float64 (1, H*W//4//blk**2), blk = 4 or 8.
This is the codec tests/detect.py constructs.  The read-out is channel 1's whatever ``scales`` says (the
reference returns wm_bits[1], dwt_dct_svd_decoder.py:24): zeros when scales[1] <= 0.  No CPU fallback."""
import numpy as np

from ..embed.dwt_dct_svd_encoder import _check_scales
from ..engine import DctEngine


class DwtDctSvdDecoder:
    def __init__(self, key=None, scales=[0, 15, 0], blk=4):
        self.key = key
        self.scales = scales
        self.blk = blk
        self._scales = _check_scales(scales, blk, need_a_mark=False)
        self._engine = None

    @property
    def engine(self) -> DctEngine:
        if self._engine is None:
            self._engine = DctEngine()
        return self._engine

    def decode(self, yuv):
        if yuv.dtype != np.float32 or yuv.ndim != 3 or yuv.shape[2] != 3:
            raise ValueError("decode expects a float32 (H, W, 3) YUV array")
        t = self.engine.torch
        dev = t.from_numpy(np.ascontiguousarray(yuv)).to(self.engine.device).unsqueeze(0)
        bits = self.engine.svd_decode_yuv(dev, scales=self._scales, blk=self.blk)
        self.block_num = bits.shape[1]
        return bits.cpu().numpy().astype(np.float64).reshape(1, -1)

    def bits_per_frame(self, height, width):
        """Length of decode()'s bit vector (dwt_dct_svd_decoder.py:14: row*col//4//blk**2): H*W//64 for blk = 4, H*W//256
        for blk = 8.  The degenerator's means divide by the slice lengths of a vector of THIS length."""
        return DctEngine.svd_bits_per_frame(height, width, self.blk)

    def decode_frames_u8(self, frames, payload_len, want_bits=False):
        """frames: CUDA uint8 [n, H, W, 3] -> (counts int32 [n, L] on device, bits or None)."""
        return self.engine.svd_detect(frames, payload_len, scales=self._scales, want_bits=want_bits, blk=self.blk)
