"""Frame sinks.  Mirrors offmark.video.frame_writer (reference src/offmark/video/frame_writer.py):
``FrameWriter.write(frame)``, ``close()``; ``FileEncoder(file, width, height)``.
Extensions the GPU pipeline (offmark.video.pipeline) uses when present: ``write_batch``; ``reserve(n)`` / ``commit(n)``
hand out page-locked memory the marked frames are DMA-written into; ``pix_fmt`` ("rgb24", "yuv420p", "nv12")."""
import logging
import shutil
import subprocess

import numpy as np

from ..common.__logging import trace

logger = logging.getLogger(__name__)


class FrameWriter:
    def __init__(self):
        pass

    def write(self, frame: np.ndarray):
        pass

    def write_batch(self, frames):
        for f in frames:
            self.write(f)

    def close(self):
        pass


class ArrayFrameWriter(FrameWriter):
    """Collects written frames in ``self.frames`` (list of per-frame uint8 arrays).

    With ``capacity`` (a frame count) and ``frame_shape`` the writer owns one page-locked block [capacity, ...] and
    hands slices of it to the pipeline (reserve / commit): marked frames land there by DMA with no host copy;
    ``array()`` is the block's filled part.  Without them every written batch is copied once."""

    def __init__(self, pix_fmt="rgb24", capacity=None, frame_shape=None):
        super().__init__()
        self.frames = []
        self.pix_fmt = pix_fmt
        self.closed = False
        self._block = None
        self._reserved = self._committed = self._pending = 0
        if capacity is not None:
            if frame_shape is None:
                raise ValueError("capacity needs frame_shape (the per-frame array shape)")
            from .pipeline import pinned_empty
            self._block = pinned_empty((int(capacity),) + tuple(frame_shape))

    def write(self, frame):
        self.frames.append(np.asarray(frame).astype(np.uint8))

    def write_batch(self, frames):
        from .pipeline import host_copy
        frames = np.asarray(frames)
        if frames.dtype != np.uint8:
            frames = frames.astype(np.uint8)             # the reference's writer casts (frame_writer.py:41-44); ADVICE r3
        kept = np.empty(frames.shape, dtype=np.uint8)    # one copy: the caller may reuse its buffer
        host_copy(kept, frames)                          # (threaded: first touch of fresh pages is what costs)
        self.frames.extend(kept)

    def reserve(self, n):
        """Page-locked room for the next n frames, or None (no block, or it is full: the caller falls back to
        write_batch)."""
        if self._block is None or self._reserved + n > len(self._block):
            return None
        view = self._block[self._reserved: self._reserved + n]
        self._reserved += n
        self._pending += n
        return view

    def commit(self, n):
        """The oldest n reserved frames have been filled."""
        self.frames.extend(self._block[self._committed: self._committed + n])
        self._committed += n
        self._pending -= n

    def array(self):
        """Every frame written so far as one [n, ...] array: a view of the page-locked block when all of them went through
        reserve/commit, else (the block was full or absent and frames arrived through write / write_batch) a stacked copy of
        ``frames`` -- never a silently shortened stream."""
        if self._block is not None and len(self.frames) == self._committed:
            return self._block[: self._committed]
        if not self.frames:
            return np.empty((0,) + (tuple(self._block.shape[1:]) if self._block is not None else ()), dtype=np.uint8)
        return np.stack(self.frames)

    def close(self):
        self.closed = True


class FileEncoder(FrameWriter):
    """Pipe rgb24 frames into an ``ffmpeg`` child that writes a yuv420p file
    (reference frame_writer.py:23-50).  Needs the ffmpeg binary on PATH."""

    def __init__(self, file, width, height, pix_fmt="rgb24"):
        super().__init__()
        self.file = file
        self.pix_fmt = pix_fmt                   # what write() receives: "rgb24" as upstream, or the 4:2:0 planes
        if not shutil.which("ffmpeg"):
            raise RuntimeError("FileEncoder needs the ffmpeg binary on PATH; use ArrayFrameWriter instead")
        self.__start_ffmpeg(width, height)

    @trace(logger)
    def __start_ffmpeg(self, width, height):
        self.ffmpeg = subprocess.Popen(["ffmpeg", "-loglevel", "quiet", "-y", "-f", "rawvideo", "-pix_fmt", self.pix_fmt,
                                        "-s", f"{width}x{height}", "-i", "pipe:", "-pix_fmt", "yuv420p", self.file],
                                       stdin=subprocess.PIPE)

    def write(self, frame):
        self.ffmpeg.stdin.write(memoryview(np.ascontiguousarray(frame, dtype=np.uint8)).cast("B"))

    def write_batch(self, frames):
        self.ffmpeg.stdin.write(memoryview(np.ascontiguousarray(frames, dtype=np.uint8)).cast("B"))

    @trace(logger)
    def close(self):
        logger.info("Waiting for ffmpeg encoder")
        self.ffmpeg.stdin.close()
        self.ffmpeg.wait()
