"""Frame sinks.  Mirrors offmark.video.frame_writer (reference src/offmark/video/frame_writer.py):
``FrameWriter.write(frame)``, ``close()``; ``FileEncoder(file, width, height)``."""
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
    """Collects written frames in ``self.frames`` (list of HxWx3 uint8)."""

    def __init__(self):
        super().__init__()
        self.frames = []
        self.closed = False

    def write(self, frame):
        self.frames.append(np.asarray(frame).astype(np.uint8))

    def write_batch(self, frames):
        kept = np.array(frames, dtype=np.uint8)          # one copy: the caller may reuse its buffer
        self.frames.extend(kept)

    def close(self):
        self.closed = True


class FileEncoder(FrameWriter):
    """Pipe rgb24 frames into an ``ffmpeg`` child that writes a yuv420p file
    (reference frame_writer.py:23-50).  Needs the ffmpeg binary on PATH."""

    def __init__(self, file, width, height):
        super().__init__()
        self.file = file
        if not shutil.which("ffmpeg"):
            raise RuntimeError("FileEncoder needs the ffmpeg binary on PATH; use ArrayFrameWriter instead")
        self.__start_ffmpeg(width, height)

    @trace(logger)
    def __start_ffmpeg(self, width, height):
        self.ffmpeg = subprocess.Popen(["ffmpeg", "-loglevel", "quiet", "-y", "-f", "rawvideo", "-pix_fmt", "rgb24",
                                        "-s", f"{width}x{height}", "-i", "pipe:", "-pix_fmt", "yuv420p", self.file],
                                       stdin=subprocess.PIPE)

    def write(self, frame):
        self.ffmpeg.stdin.write(frame.astype(np.uint8).tobytes())

    @trace(logger)
    def close(self):
        logger.info("Waiting for ffmpeg encoder")
        self.ffmpeg.stdin.close()
        self.ffmpeg.wait()
