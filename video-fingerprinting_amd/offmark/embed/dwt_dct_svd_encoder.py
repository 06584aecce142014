"""DwtDctSvdEncoder on the MI355X.  Mirrors offmark.embed.dwt_dct_svd_encoder.DwtDctSvdEncoder
(reference src/offmark/embed/dwt_dct_svd_encoder.py:5-45): DwtDctSvdEncoder(key=None,
scales=[0,15,0], blk=4), read_wm(wm), wm_capacity(frame_shape), encode(yuv) (mutates and returns).
This is the codec tests/mark.py constructs.  Any per-channel ``scales`` are supported (every channel with a
positive scale is marked with the same watermark, dwt_dct_svd_encoder.py:19-26).  ``blk`` is 4 (the default: 8x8
pixel tiles, one watermark bit per tile) or 8 (16x16 pixel tiles: tile c takes wm[c], so the first quarter of the
(1, row*col//64) watermark is used, dwt_dct_svd_encoder.py:29-40).  Smaller blocks index past the reference's own
watermark (its capacity stays row*col//64, dwt_dct_svd_encoder.py:14-17); larger ones are not built.
No CPU fallback."""
import numpy as np

from ..engine import DctEngine


def _check_scales(scales, blk, need_a_mark=True):
    scales = [float(x) for x in scales]
    if len(scales) != 3:
        raise ValueError("scales needs three entries (one per YUV channel); got %r" % (scales,))
    if blk not in (4, 8):
        raise NotImplementedError("the HIP DwtDctSvd codec implements blk=4 (the reference's default) and blk=8; smaller blocks "
                                  "index past the reference's own watermark, larger ones are not built; got blk=%r" % (blk,))
    for x in scales:
        # the same rule as the C ABI (offmark_kernels.hip: set_scales): decided on the float32 value the kernels use
        if not np.isfinite(x) or (x > 0 and not np.float32(x) >= np.float32(1e-3)):
            raise ValueError("scales must be finite, and a positive scale at least 1e-3 as float32; got %r" % (scales,))
    if need_a_mark and not any(x > 0 for x in scales):
        raise ValueError("no channel has a positive scale: nothing would be marked")
    return scales


class DwtDctSvdEncoder:
    def __init__(self, key=None, scales=[0, 15, 0], blk=4):
        self.key = key
        self.scales = scales
        self.blk = blk
        self._scales = _check_scales(scales, blk)
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
        self.engine.svd_encode_yuv(dev, self._device_wm(h * w // 64), scales=self._scales, blk=self.blk)
        if yuv.flags.c_contiguous and yuv.flags.writeable:
            # one contiguous download into the caller's array: the kernels write the marked channels' covered tiles only
            # (csrc/svd_kernels.hiph, svd8_kernels.hiph), every other float comes back as the bits that went up
            t.from_numpy(yuv).copy_(dev[0])
        else:
            back = dev[0].cpu().numpy()
            for ch in range(3):
                if self._scales[ch] > 0:
                    yuv[:, :, ch] = back[:, :, ch]
        return yuv

    def encode_frames_u8(self, frames, out=None, wm_rows=None, wm_table=None):
        """frames: CUDA uint8 [n, H, W, 3]: the whole reference frame step (embedder.py:33-39) on device."""
        n, h, w, _ = frames.shape
        wm = wm_table if wm_table is not None else self._device_wm(h * w // 64)
        return self.engine.svd_embed(frames, wm, scales=self._scales, wm_row=wm_rows, out=out, blk=self.blk)
