"""DctEncoder on the MI355X.  Mirrors offmark.embed.dct_encoder.DctEncoder
(reference src/offmark/embed/dct_encoder.py:4-102): DctEncoder(key=None, alpha=20),
read_wm(wm), wm_capacity(frame_shape), encode(yuv) (mutates and returns its argument),
luminance_mask(lum), texture_mask(lum).

``encode(yuv)`` is the literal plugin boundary (float32 YUV ndarray, one frame); it uploads the
frame, runs the HIP kernels and writes channel 1 back in place.  The fast path is
``encode_frames_u8`` (device u8 RGB batches, no YUV round trip through memory), which
``offmark.video.embedder.Embedder`` uses when it sees this class.  No CPU fallback.
"""
import numpy as np

from ..engine import DctEngine


class DctEncoder:
    def __init__(self, key=None, alpha=20):
        self.key = key          # unused by the reference too (dct_encoder.py:6-8)
        self.alpha = alpha
        self.wm = None
        self._engine = None
        self._wm_dev = None

    @property
    def engine(self) -> DctEngine:
        if self._engine is None:
            self._engine = DctEngine()
        return self._engine

    def read_wm(self, wm):
        self.wm = np.asarray(wm)[0]
        self._wm_dev = None

    def wm_capacity(self, frame_shape):
        row, col, _channels = frame_shape
        return (1, row * col // 64)

    def _device_wm(self, n_bits):
        if self.wm is None:
            raise RuntimeError("read_wm() must be called before encode()")
        if self.wm.size < n_bits:
            raise ValueError(f"watermark has {self.wm.size} bits, frame needs {n_bits}")
        if self._wm_dev is None or self._wm_dev.shape[1] != n_bits:
            t = self.engine.torch
            self._wm_dev = t.from_numpy((self.wm[:n_bits] != 0).astype(np.uint8)).reshape(1, n_bits).to(self.engine.device)
        return self._wm_dev

    # -- plugin API ------------------------------------------------------------------------------
    def encode(self, yuv):
        if yuv.dtype != np.float32 or yuv.ndim != 3 or yuv.shape[2] != 3:
            raise ValueError("encode expects a float32 (H, W, 3) YUV array")
        t = self.engine.torch
        h, w, _ = yuv.shape
        dev = t.from_numpy(np.ascontiguousarray(yuv)).to(self.engine.device).unsqueeze(0)
        self.engine.encode_yuv(dev, self._device_wm(h * w // 64), alpha=self.alpha)
        if yuv.flags.c_contiguous and yuv.flags.writeable:
            # ONE contiguous download straight into the caller's array.  The kernels write channel 1 only (dct_encoder.py:20,36-37),
            # so channels 0 and 2 come back as the very bits that went up: identical to writing channel 1 alone, without the strided
            # device slice + strided host write that made this call 9x slower than decode
            t.from_numpy(yuv).copy_(dev[0])
        else:
            yuv[:, :, 1] = dev[0, :, :, 1].contiguous().cpu().numpy()
        return yuv

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
    def encode_frames_u8(self, frames, out=None, wm_rows=None, wm_table=None):
        """frames: CUDA uint8 [n, H, W, 3].  Whole reference frame step (embedder.py:33-39) on device."""
        n, h, w, _ = frames.shape
        wm = wm_table if wm_table is not None else self._device_wm(h * w // 64)
        return self.engine.embed(frames, wm, alpha=self.alpha, wm_row=wm_rows, out=out)

    def encode_planes_yuv420(self, planes, height, width, out=None, wm_rows=None, wm_table=None, layout="i420"):
        """planes: CUDA uint8 [n, 1.5*H*W] (I420: Y|U|V per frame, NV12: Y|UV): the frame step on what a decoder produces
        and an encoder takes (reference: ffmpeg's rgb24 pipe and yuv420p writer, frame_reader.py:42-64, frame_writer.py:33-34),
        with the build-defined BT.601 conversion fused into the kernels.  Returns marked planes of the same layout."""
        wm = wm_table if wm_table is not None else self._device_wm(height * width // 64)
        return self.engine.embed_yuv420(planes, height, width, wm, alpha=self.alpha, wm_row=wm_rows, out=out, layout=layout)
