"""Extractor pipeline.  Mirrors offmark.video.extractor.Extractor
(reference src/offmark/video/extractor.py:11-34): ``Extractor(frame_reader, frame_extractor,
degenerator).start()`` logs each frame's recovered payload at INFO.

Extension: the payloads are also kept in ``self.patterns`` (the reference's
PatternCollectorExtractor, tests/segment_mark_detect_hls.py:119-161, re-implements the loop just to
collect them) and ``most_common()`` gives that class's (pattern, frequency) vote.

Batched GPU path when ``frame_extractor`` offers ``decode_frames_u8`` and the degenerator offers
``degenerate_counts`` (HIP DctDecoder + DeShuffler/DeGrayScale); generic per-frame path otherwise.
"""
import logging

import numpy as np

from ..common.__logging import trace
from ..dist.vote import vote
from .color import bgr2yuv

logger = logging.getLogger(__name__)


class Extractor:
    def __init__(self, frame_reader, frame_extractor, degenerator, batch_frames=64):
        self.frame_reader = frame_reader
        self.frame_extractor = frame_extractor
        self.degenerator = degenerator
        self.batch_frames = batch_frames
        self.patterns = []

    @trace(logger)
    def start(self):
        batched = hasattr(self.frame_extractor, "decode_frames_u8") and hasattr(self.degenerator, "degenerate_counts")
        if batched:
            self.__run_batched()
        else:
            while True:
                in_frame = self.frame_reader.read()
                if in_frame is None:
                    logger.info("End of input stream")
                    break
                self.__check_frame(in_frame)
        self.frame_reader.close()
        logger.info("Done")

    def most_common(self):
        """(most common whole pattern, its frequency) over the frames seen, or (None, None)."""
        flat = [np.asarray(p).reshape(-1) for p in self.patterns]
        return vote(np.stack(flat)) if flat else (None, None)

    def __run_batched(self):
        import torch
        L = self.degenerator.payload_len
        while True:
            batch = self.__next_batch()
            if batch is None:
                logger.info("End of input stream")
                break
            n, h, w, _ = batch.shape
            dev = torch.from_numpy(np.ascontiguousarray(batch)).to(self.frame_extractor.engine.device)
            counts, _ = self.frame_extractor.decode_frames_u8(dev, L)
            outs = self.degenerator.degenerate_counts(counts.cpu().numpy(), h * w // 64)
            for out in outs:
                self.patterns.append(out)
                logger.info(out)

    def __next_batch(self):
        if hasattr(self.frame_reader, "read_batch"):
            return self.frame_reader.read_batch(self.batch_frames)
        frames = []                                   # duck-typed reader with only read()/close()
        while len(frames) < self.batch_frames:
            f = self.frame_reader.read()
            if f is None:
                break
            frames.append(f)
        return np.stack(frames) if frames else None

    def __check_frame(self, frame_rgb):
        wm_frame_yuv = bgr2yuv(frame_rgb.astype(np.float32))
        frame_yuv = self.frame_extractor.decode(wm_frame_yuv)
        out = self.degenerator.degenerate(frame_yuv)
        self.patterns.append(out)
        logger.info(out)
