"""Embedder pipeline.  Mirrors offmark.video.embedder.Embedder
(reference src/offmark/video/embedder.py:11-39): ``Embedder(frame_reader, frame_embedder,
frame_writer).start()`` reads until ``read()`` returns None, marks, writes, closes both ends.

Two paths:
  * batched GPU path, when ``frame_embedder`` offers ``encode_frames_u8`` (the HIP codecs):
    up to ``batch_frames`` frames per launch go to the device as u8 (RGB, or 4:2:0 planes when the
    reader says so), the whole reference frame step (embedder.py:33-39) runs in the kernels, marked u8
    frames come back -- pipelined over three streams, see offmark.video.pipeline;
  * generic path for any other duck-typed encoder: the reference's per-frame sequence
    u8 -> f32 -> BGR2YUV -> encode(yuv) -> YUV2BGR -> clip -> around -> u8 on the host.
"""
import logging

import numpy as np

from ..common.__logging import trace
from .color import bgr2yuv, yuv2bgr

logger = logging.getLogger(__name__)


class Embedder:
    def __init__(self, frame_reader, frame_embedder, frame_writer, batch_frames=64):
        self.frame_reader = frame_reader
        self.frame_writer = frame_writer
        self.frame_embedder = frame_embedder
        self.batch_frames = batch_frames
        self.frames_marked = 0

    @trace(logger)
    def start(self):
        if hasattr(self.frame_embedder, "encode_frames_u8"):
            self.__run_batched()
        else:
            self.__run_per_frame()
        self.frame_reader.close()
        self.frame_writer.close()
        logger.info("Done")

    def __run_per_frame(self):
        while True:
            in_frame = self.frame_reader.read()
            if in_frame is None:
                logger.info("End of input stream")
                break
            self.frame_writer.write(self.__mark_frame(in_frame))
            self.frames_marked += 1

    def __run_batched(self):
        """The frame loop as a three-stream pipeline (offmark.video.pipeline): upload, kernels and download of
        neighbouring batches overlap, the reader is read ahead on a helper thread, and page-locked memory is used at
        both ends -- the reader's / writer's own when they offer it, staging buffers otherwise.  Readers and
        writers may carry ``pix_fmt`` "yuv420p" / "nv12": the planes cross PCIe (half the bytes) and the conversion
        is fused into the kernels (or done on the device when the codec has no planar kernels)."""
        from . import pipeline as pl
        enc = self.frame_embedder
        eng = enc.engine
        reader = self.frame_reader
        if not (hasattr(reader, "height") and hasattr(reader, "width")):
            reader = pl.PeekedReader(reader)
            if reader.first is None:
                logger.info("End of input stream")
                return
        H, W = int(reader.height), int(reader.width)
        in_fmt, out_fmt = pl.pix_fmt_of(reader), pl.pix_fmt_of(self.frame_writer)
        fused_planar = in_fmt == out_fmt != "rgb24" and hasattr(enc, "encode_planes_yuv420")

        def process(dev_in, dev_out):
            m = dev_in.shape[0]
            if fused_planar:
                enc.encode_planes_yuv420(dev_in.view(m, -1), H, W, out=dev_out.view(m, -1), layout=pl.PLANAR_LAYOUT[in_fmt])
                return
            rgb = pl.to_rgb_on_device(eng, dev_in, in_fmt, H, W)
            if out_fmt == "rgb24":
                enc.encode_frames_u8(rgb, out=dev_out)
            else:
                eng.rgb_to_yuv420(enc.encode_frames_u8(rgb), layout=pl.PLANAR_LAYOUT[out_fmt], out=dev_out.view(m, -1))

        pipe = pl.StagedPipeline(eng.device, reader, pl.batch_size(self.batch_frames, pl.frame_shape(in_fmt, H, W)),
                                 pl.frame_shape(in_fmt, H, W), pl.frame_shape(out_fmt, H, W), np.uint8)
        try:
            pipe.run(process, pl.WriterSink(self.frame_writer))
        finally:
            self.frames_marked += pipe.frames_done
        logger.info("End of input stream")

    def __mark_frame(self, frame_rgb):
        frame_yuv = bgr2yuv(frame_rgb.astype(np.float32))
        wm_frame_yuv = self.frame_embedder.encode(frame_yuv)
        wm_frame_rgb = np.clip(yuv2bgr(wm_frame_yuv), a_min=0, a_max=255)
        return np.around(wm_frame_rgb).astype(np.uint8)
