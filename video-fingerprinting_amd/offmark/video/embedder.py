"""Embedder pipeline.  Mirrors offmark.video.embedder.Embedder
(reference src/offmark/video/embedder.py:11-39): ``Embedder(frame_reader, frame_embedder,
frame_writer).start()`` reads until ``read()`` returns None, marks, writes, closes both ends.

Two paths:
  * batched GPU path, when ``frame_embedder`` offers ``encode_frames_u8`` (the HIP DctEncoder):
    up to ``batch_frames`` frames per launch go to the device as u8 RGB, the whole reference
    frame step (embedder.py:33-39) runs in the kernels, marked u8 frames come back;
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
        """Two batches in flight: while batch k runs on the GPU and drains into a pinned host buffer, batch
        k-1 is handed to the writer.  (Frames cross PCIe twice here; bench.py measures HBM-resident frames.)"""
        import torch
        dev = self.frame_embedder.engine.device
        read_batch = getattr(self.frame_reader, "read_batch", None)
        pinned, done, pending = [None, None], [torch.cuda.Event(), torch.cuda.Event()], None
        k = 0
        while True:
            batch = read_batch(self.batch_frames) if read_batch else self.__collect()
            if batch is not None:
                slot = k & 1
                src = torch.from_numpy(np.ascontiguousarray(batch))
                if pinned[slot] is None or pinned[slot].shape[1:] != src.shape[1:] or pinned[slot].shape[0] < src.shape[0]:
                    pinned[slot] = torch.empty((self.batch_frames,) + tuple(src.shape[1:]), dtype=torch.uint8).pin_memory()
                marked = self.frame_embedder.encode_frames_u8(src.to(dev, non_blocking=True))
                pinned[slot][: len(src)].copy_(marked, non_blocking=True)
                done[slot].record()
            if pending is not None:
                slot, count = pending
                done[slot].synchronize()
                self.__write_all(pinned[slot][:count].numpy())
                self.frames_marked += count
            if batch is None:
                logger.info("End of input stream")
                break
            pending = (k & 1, len(batch))
            k += 1

    def __write_all(self, frames):
        if hasattr(self.frame_writer, "write_batch"):
            self.frame_writer.write_batch(frames)
        else:
            for f in frames:
                self.frame_writer.write(f)

    def __collect(self):
        frames = []
        while len(frames) < self.batch_frames:
            f = self.frame_reader.read()
            if f is None:
                break
            frames.append(f)
        return np.stack(frames) if frames else None

    def __mark_frame(self, frame_rgb):
        frame_yuv = bgr2yuv(frame_rgb.astype(np.float32))
        wm_frame_yuv = self.frame_embedder.encode(frame_yuv)
        wm_frame_rgb = np.clip(yuv2bgr(wm_frame_yuv), a_min=0, a_max=255)
        return np.around(wm_frame_rgb).astype(np.uint8)
