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
        """The same three-stream pipeline as the Embedder's (offmark.video.pipeline), two batches in flight: while
        batch k's frames upload and its kernels run, batch k-1's counts come back and are degenerated on the host.
        Only [n, L] int32 counts travel device -> host."""
        from . import pipeline as pl
        dec = self.frame_extractor
        eng = dec.engine
        L = self.degenerator.payload_len
        reader = self.frame_reader
        if not (hasattr(reader, "height") and hasattr(reader, "width")):
            reader = pl.PeekedReader(reader)
            if reader.first is None:
                logger.info("End of input stream")
                return
        H, W = int(reader.height), int(reader.width)
        fmt = pl.pix_fmt_of(reader)
        planar = fmt != "rgb24" and hasattr(dec, "decode_planes_yuv420")

        def process(dev_in, dev_out):
            if planar:
                counts, _ = dec.decode_planes_yuv420(dev_in.view(dev_in.shape[0], -1), H, W, L, layout=pl.PLANAR_LAYOUT[fmt])
            else:
                counts, _ = dec.decode_frames_u8(pl.to_rgb_on_device(eng, dev_in, fmt, H, W), L)
            dev_out.copy_(counts)

        outer = self
        # length of the decoder's bit vector: H*W//64 for the DCT codec and DwtDctSvd(blk=4), H*W//256 for DwtDctSvd(blk=8)
        n_bits = dec.bits_per_frame(H, W) if hasattr(dec, "bits_per_frame") else H * W // 64

        class Sink:
            def deliver(self, counts):
                for out in outer.degenerator.degenerate_counts(counts, n_bits):
                    outer.patterns.append(out)
                    logger.info(out)

        pl.StagedPipeline(eng.device, reader, pl.batch_size(self.batch_frames, pl.frame_shape(fmt, H, W)), pl.frame_shape(fmt, H, W),
                          (L,), np.int32).run(process, Sink())
        logger.info("End of input stream")

    def __check_frame(self, frame_rgb):
        wm_frame_yuv = bgr2yuv(frame_rgb.astype(np.float32))
        frame_yuv = self.frame_extractor.decode(wm_frame_yuv)
        out = self.degenerator.degenerate(frame_yuv)
        self.patterns.append(out)
        logger.info(out)
