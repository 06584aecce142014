"""DctDecoder on the MI355X.  Mirrors offmark.extract.dct_decoder.DctDecoder
(reference src/offmark/extract/dct_decoder.py:4-89): DctDecoder(key=None, alpha=20),
decode(yuv) -> float64 array (1, H*W//64) of 0./1., luminance_mask, texture_mask.  No CPU fallback."""
import numpy as np

from ..engine import DctEngine


class DctDecoder:
    def __init__(self, key=None, alpha=20):
        self.key = key
        self.alpha = alpha
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
        _counts, bits = self.engine.decode_yuv(dev, L=1, alpha=self.alpha, want_bits=True)
        return bits.cpu().numpy().astype(np.float64).reshape(1, -1)

    def _planes(self, lum):
        t = self.engine.torch
        lum = np.ascontiguousarray(lum, dtype=np.float32)
        yuv = np.zeros(lum.shape + (3,), np.float32)
        yuv[:, :, 0] = lum
        return self.engine.debug_planes(t.from_numpy(yuv).to(self.engine.device), alpha=self.alpha)

    def luminance_mask(self, lum):
        return self._planes(lum)["lum"]

    def texture_mask(self, lum):
        return self._planes(lum)["tex"]

    # -- batch fast path ---------------------------------------------------------------------------
    def bits_per_frame(self, height, width):
        """Length of decode()'s bit vector for a frame of this size (dct_decoder.py:16): what the degenerator's means divide by."""
        return height * width // 64

    def decode_frames_u8(self, frames, payload_len, want_bits=False):
        """frames: CUDA uint8 [n, H, W, 3] -> (counts int32 [n, L] on device, bits or None)."""
        return self.engine.detect(frames, payload_len, alpha=self.alpha, want_bits=want_bits)

    def decode_planes_yuv420(self, planes, height, width, payload_len, want_bits=False, layout="i420"):
        """planes: CUDA uint8 [n, 1.5*H*W] (I420 or NV12) -> (counts int32 [n, L] on device, bits or None)."""
        return self.engine.detect_yuv420(planes, height, width, payload_len, alpha=self.alpha, want_bits=want_bits, layout=layout)
