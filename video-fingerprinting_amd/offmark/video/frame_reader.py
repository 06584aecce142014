"""Frame sources.  Mirrors offmark.video.frame_reader (reference src/offmark/video/frame_reader.py):
``FrameReader.read() -> HxWx3 uint8 | None``, ``close()``; ``FileDecoder(file)`` with
``width`` / ``height``.

ArrayFrameReader is the in-memory source (tests, benches, callers that decode elsewhere).
Extensions the GPU pipeline (offmark.video.pipeline) uses when present: ``read_batch`` pulls several frames per
launch, ``read_batch_into`` fills the pipeline's page-locked buffer directly, ``pinned`` says that batches are views
of page-locked memory, ``pix_fmt`` names what a frame is ("rgb24" [H,W,3]; "yuv420p" / "nv12": [H*3/2, W] planes --
the reference's own open question at frame_reader.py:27)."""
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
    """Frames from an array [n, ...] uint8 (or any sequence of per-frame arrays): [H, W, 3] for rgb24,
    [H*3/2, W] for the planar 4:2:0 formats.

    pin=True page-locks a contiguous array in place (hipHostRegister) so that batches are DMA sources as they are;
    pin="already" promises the memory is page-locked (e.g. a pinned torch tensor's .numpy())."""

    def __init__(self, frames, pix_fmt="rgb24", pin=False):
        super().__init__()
        self.frames = frames
        self.pix_fmt = pix_fmt
        self.pos = 0
        first = np.asarray(frames[0]).shape
        self.height, self.width = (first[0], first[1]) if pix_fmt == "rgb24" else (first[0] * 2 // 3, first[1])
        self.closed = False
        self.pinned = False
        self._registration = None
        if pin:
            if not (isinstance(frames, np.ndarray) and frames.flags.c_contiguous):
                raise ValueError("pin needs one C-contiguous ndarray of frames")
            if pin != "already":
                from .pipeline import HostRegistration
                self._registration = HostRegistration(frames)
            self.pinned = True

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
        if self._registration is not None:
            self._registration.close()
            self._registration = None
            self.pinned = False


class FileDecoder(FrameReader):
    """Decode a video file to rgb24 frames through an ``ffmpeg`` child process
    (reference frame_reader.py:28-69).  Needs the ffmpeg/ffprobe binaries on PATH."""

    def __init__(self, file, pix_fmt="rgb24"):
        super().__init__()
        self.file = file
        self.pix_fmt = pix_fmt                    # "rgb24" as upstream; "yuv420p" / "nv12": ffmpeg hands over the planes
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
        from .pipeline import frame_shape
        self.frame_shape = frame_shape(self.pix_fmt, self.height, self.width)
        self.frame_size_bytes = int(np.prod(self.frame_shape))
        self.ffmpeg = subprocess.Popen(["ffmpeg", "-loglevel", "quiet", "-i", self.file, "-f", "rawvideo",
                                        "-pix_fmt", self.pix_fmt, "pipe:"], stdout=subprocess.PIPE)

    def read(self):
        data = self.ffmpeg.stdout.read(self.frame_size_bytes)
        if len(data) == 0:
            return None
        assert len(data) == self.frame_size_bytes
        return np.frombuffer(data, np.uint8).reshape(self.frame_shape)

    def read_batch_into(self, buf):
        """Fill ``buf`` [B, ...] uint8 (the pipeline's page-locked staging) straight from the pipe; -> frames read."""
        flat = memoryview(buf).cast("B")
        want, got = len(flat), 0
        while got < want:
            k = self.ffmpeg.stdout.readinto(flat[got:])
            if not k:
                break
            got += k
        assert got % self.frame_size_bytes == 0, "ffmpeg ended inside a frame"
        return got // self.frame_size_bytes

    @trace(logger)
    def close(self):
        logger.info("Waiting for ffmpeg decoder")
        self.ffmpeg.wait()
