"""DwtDctSvdEncoder on the MI355X.  Mirrors offmark.embed.dwt_dct_svd_encoder.DwtDctSvdEncoder
(reference src/offmark/embed/dwt_dct_svd_encoder.py:5-45): DwtDctSvdEncoder(key=None,
scales=[0,15,0], blk=4), read_wm(wm), wm_capacity(frame_shape), encode(yuv) (mutates and returns).
This is the codec tests/mark.py constructs.  Supported configuration: the reference's default shape,
i.e. only channel 1 carries a mark (scales = [0, s, 0]) and blk = 4.  No CPU fallback."""
import numpy as np

from ..engine import DctEngine


def _single_scale(scales, blk):
    scales = list(scales)
    if blk != 4 or len(scales) != 3 or scales[0] > 0 or scales[2] > 0 or not scales[1] > 0:
        raise NotImplementedError("the HIP DwtDctSvd codec supports scales=[0, s, 0] with s > 0 and blk=4 "
                                  "(the reference's defaults); got scales=%r blk=%r" % (scales, blk))
    return scales[1]


class DwtDctSvdEncoder:
    def __init__(self, key=None, scales=[0, 15, 0], blk=4):
        self.key = key
        self.scales = scales
        self.blk = blk
        self._scale = _single_scale(scales, blk)
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

    def encode(self, yuv):
        if yuv.dtype != np.float32 or yuv.ndim != 3 or yuv.shape[2] != 3:
            raise ValueError("encode expects a float32 (H, W, 3) YUV array")
        t = self.engine.torch
        h, w, _ = yuv.shape
        dev = t.from_numpy(np.ascontiguousarray(yuv)).to(self.engine.device).unsqueeze(0)
        self.engine.svd_encode_yuv(dev, self._device_wm(h * w // 64), scale=self._scale)
        yuv[:, :, 1] = dev[0, :, :, 1].cpu().numpy()
        return yuv

    def encode_frames_u8(self, frames, out=None, wm_rows=None, wm_table=None):
        """frames: CUDA uint8 [n, H, W, 3]: the whole reference frame step (embedder.py:33-39) on device."""
        n, h, w, _ = frames.shape
        wm = wm_table if wm_table is not None else self._device_wm(h * w // 64)
        return self.engine.svd_embed(frames, wm, scale=self._scale, wm_row=wm_rows, out=out)
