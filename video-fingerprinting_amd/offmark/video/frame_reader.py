"""Frame sources.  Mirrors offmark.video.frame_reader (reference src/offmark/video/frame_reader.py):
``FrameReader.read() -> HxWx3 uint8 | None``, ``close()``; ``FileDecoder(file)`` with
``width`` / ``height``.

ArrayFrameReader is the in-memory source (tests, benches, callers that decode elsewhere).
``read_batch`` is an extension the GPU pipeline uses to pull several frames per launch."""
import json
import logging
import shutil
import subprocess

import numpy as np

from ..common.__logging import trace

logger = logging.getLogger(__name__)


class FrameReader:
    def __init__(self):
        pass

    def read(self) -> np.ndarray:
        """Read one frame in RGB format."""
        pass

    def read_batch(self, max_frames: int):
        """Up to ``max_frames`` frames stacked as [n, H, W, 3], or None at end of stream."""
        frames = []
        while len(frames) < max_frames:
            f = self.read()
            if f is None:
                break
            frames.append(f)
        return np.stack(frames) if frames else None

    def close(self):
        pass


class ArrayFrameReader(FrameReader):
    """Frames from an array [n, H, W, 3] uint8 (or any sequence of HxWx3 arrays)."""

    def __init__(self, frames):
        super().__init__()
        self.frames = frames
        self.pos = 0
        self.height, self.width = np.asarray(frames[0]).shape[:2]
        self.closed = False

    def read(self):
        if self.pos >= len(self.frames):
            return None
        f = np.asarray(self.frames[self.pos])
        self.pos += 1
        return f

    def read_batch(self, max_frames):
        if self.pos >= len(self.frames):
            return None
        end = min(len(self.frames), self.pos + max_frames)
        out = np.asarray(self.frames[self.pos:end])
        self.pos = end
        return out

    def close(self):
        self.closed = True


class FileDecoder(FrameReader):
    """Decode a video file to rgb24 frames through an ``ffmpeg`` child process
    (reference frame_reader.py:28-69).  Needs the ffmpeg/ffprobe binaries on PATH."""

    def __init__(self, file):
        super().__init__()
        self.file = file
        if not (shutil.which("ffmpeg") and shutil.which("ffprobe")):
            raise RuntimeError("FileDecoder needs the ffmpeg and ffprobe binaries on PATH; "
                               "use ArrayFrameReader for frames that are already decoded")
        self.__start_ffmpeg()

    @trace(logger)
    def __start_ffmpeg(self):
        probe = subprocess.run(["ffprobe", "-v", "quiet", "-print_format", "json", "-show_streams", self.file],
                               check=True, capture_output=True)
        video = next(s for s in json.loads(probe.stdout)["streams"] if s["codec_type"] == "video")
        self.width, self.height = int(video["width"]), int(video["height"])
        self.frame_size_bytes = self.width * self.height * 3
        self.ffmpeg = subprocess.Popen(["ffmpeg", "-loglevel", "quiet", "-i", self.file, "-f", "rawvideo",
                                        "-pix_fmt", "rgb24", "pipe:"], stdout=subprocess.PIPE)

    def read(self):
        data = self.ffmpeg.stdout.read(self.frame_size_bytes)
        if len(data) == 0:
            return None
        assert len(data) == self.frame_size_bytes
        return np.frombuffer(data, np.uint8).reshape(self.height, self.width, 3)

    @trace(logger)
    def close(self):
        logger.info("Waiting for ffmpeg decoder")
        self.ffmpeg.wait()
