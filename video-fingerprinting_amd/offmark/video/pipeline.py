"""Host <-> GPU frame pipeline under Embedder and Extractor.

The reference reads one frame from a pipe, processes it, writes it, and only then reads the next
(src/offmark/video/embedder.py:18-31, extractor.py:18-28).  Here batches of frames flow through three
HIP streams -- upload, kernels, download -- with two device batches in flight, so that PCIe in, the
kernels and PCIe out of neighbouring batches overlap; reading the next batch from the frame reader runs
on a helper thread (host work only: no HIP call is made off the caller's thread).

What crosses the boundary, per frame (``pix_fmt`` attribute of the reader / writer, "rgb24" when absent):
  rgb24    [H, W, 3] uint8, what the reference's ffmpeg pipes carry (frame_reader.py:42-64)
  yuv420p  [H*3/2, W] uint8 = the I420 planes Y | U | V of one frame (the reference's own open question,
           frame_reader.py:27 "pix_fmt yuv420p?", and what its writer has ffmpeg produce, frame_writer.py:33-34)
  nv12     [H*3/2, W] uint8 = Y | interleaved UV
Half the bytes per frame cross PCIe with the planar formats, and the codec's planar kernels convert on the fly.

Copies the host has to make are what bounds a plugin-level pipeline once the kernels run at TB/s, so every
hand-over has a zero-copy form, all optional (duck typing, as in the reference):
  reader.pinned = True            its read_batch() results are views of page-locked memory: DMA straight from them
  reader.read_batch_into(buf)     fill the pipeline's page-locked staging buffer itself (a pipe's readinto); -> count
  reader.read_batch(n) / read()   anything else: one host copy into staging (split over a few threads)
  writer.reserve(n) / commit(n)   hand out page-locked memory for the next n frames: DMA straight into it
  writer.write_batch(a) / write() anything else: called with views of the staging buffer; a writer that keeps
                                  frames must copy them (ArrayFrameWriter does)
"""
from __future__ import annotations

import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

PLANAR_LAYOUT = {"yuv420p": "i420", "nv12": "nv12"}
PIX_FMTS = ("rgb24",) + tuple(PLANAR_LAYOUT)


def pix_fmt_of(obj) -> str:
    fmt = getattr(obj, "pix_fmt", None) or "rgb24"
    if fmt not in PIX_FMTS:
        raise ValueError(f"unsupported pix_fmt {fmt!r}: one of {PIX_FMTS}")
    return fmt


def batch_size(batch_frames: int, item_shape, limit_bytes: int = 512 << 20) -> int:
    """Frames per pipeline batch: the caller's ``batch_frames``, cut down so that one batch stays under ``limit_bytes``
    (the pipeline holds ~9 batch-sized buffers, five of them page-locked: 64 frames of 4K would be 14 GB of pinned memory)."""
    per = int(np.prod(item_shape))
    return max(1, min(int(batch_frames), limit_bytes // max(per, 1)))


def frame_shape(pix_fmt: str, height: int, width: int) -> tuple:
    """Per-frame array shape at the reader / writer boundary."""
    if pix_fmt == "rgb24":
        return (height, width, 3)
    if height % 2 or width % 2:
        raise ValueError("4:2:0 frames need even height and width")
    return (height * 3 // 2, width)


# ---- page-locked host memory -------------------------------------------------------------------------------------
def pinned_empty(shape, dtype=np.uint8) -> np.ndarray:
    """ndarray over freshly allocated page-locked memory (torch's caching host allocator owns it; the array keeps
    the tensor alive)."""
    import torch
    # allocated page-locked in the first place (not .pin_memory(): that allocates pageable memory, touches it and copies it);
    # blocks come back from torch's caching host allocator, so a second pipeline run does not pay hipHostMalloc again
    t = torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True)
    return t.numpy()


class HostRegistration:
    """Page-lock an existing C-contiguous ndarray in place (hipHostRegister) so that the copy engines can read or
    write it directly; undone by close() / garbage collection.  For callers whose frames are already in memory."""

    def __init__(self, array: np.ndarray):
        import torch
        if not array.flags.c_contiguous:
            raise ValueError("only a C-contiguous array can be page-locked in place")
        self.array = array
        self._rt = torch.cuda.cudart()
        self._ptr = array.ctypes.data
        rc = int(self._rt.cudaHostRegister(self._ptr, array.nbytes, 0)) if array.nbytes else 0
        if rc != 0:
            self._ptr = None
            raise RuntimeError(f"hipHostRegister failed with code {rc} for {array.nbytes} bytes")

    def close(self):
        if self._ptr is not None and self.array.nbytes:
            # no copy engine may still be reading or writing these pages when they stop being page-locked (the pipeline has synchronised its
            # streams by now; a caller that used the array for copies of its own may not have), and a failed unregistration must not pass in
            # silence: the range would stay registered while its memory goes back to the allocator
            import torch
            if torch.cuda.is_initialized():
                torch.cuda.synchronize()
            rc = int(self._rt.cudaHostUnregister(self._ptr))
            if rc != 0:
                import logging
                logging.getLogger(__name__).warning("hipHostUnregister failed with code %d for %d bytes at %#x", rc, self.array.nbytes, self._ptr)
        self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_COPY_THREADS = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
_pool = None


def host_copy(dst: np.ndarray, src: np.ndarray) -> None:
    """dst[...] = src for frame batches; large ones are split over a few threads (numpy drops the GIL while it copies;
    one core's memcpy is ~10 GB/s, a tenth of what the rest of this pipeline moves)."""
    global _pool
    n = len(src)
    if _COPY_THREADS == 1 or n < 2 or src.nbytes < (16 << 20):
        np.copyto(dst, src)
        return
    if _pool is None:
        _pool = ThreadPoolExecutor(_COPY_THREADS, thread_name_prefix="offmark-copy")
    cuts = np.linspace(0, n, min(_COPY_THREADS, n) + 1).astype(int)
    for f in [_pool.submit(np.copyto, dst[a:b], src[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]:
        f.result()


# ---- reading ahead ------------------------------------------------------------------------------------------------
class _ReadAhead(threading.Thread):
    """Pulls batches from the frame reader into page-locked memory while the caller's thread drives the GPU and the
    writer.  Host work only.  Items on ``ready``: (array [m, ...], staging buffer or None), None at end of stream, or
    the exception the reader raised."""

    def __init__(self, reader, batch, item_shape, dtype, staging):
        super().__init__(name="offmark-read-ahead", daemon=True)
        self.reader, self.batch, self.item_shape, self.dtype = reader, batch, tuple(item_shape), np.dtype(dtype)
        self.free = queue.Queue()
        for s in staging:
            self.free.put(s)
        self.ready = queue.Queue(maxsize=max(1, len(staging)) if staging else 2)
        self.stop = threading.Event()

    def _check(self, a):
        if tuple(a.shape[1:]) != self.item_shape or a.dtype != self.dtype:
            raise ValueError(f"frame reader delivered {a.dtype}{tuple(a.shape[1:])}, expected {self.dtype}{self.item_shape}")
        return a

    def _next(self):
        r = self.reader
        if getattr(r, "pinned", False) and hasattr(r, "read_batch"):
            b = r.read_batch(self.batch)
            return None if b is None or len(b) == 0 else (self._check(np.asarray(b)), None)
        st = self.free.get()
        if st is None:
            return None
        if hasattr(r, "read_batch_into"):
            m = int(r.read_batch_into(st))
        elif hasattr(r, "read_batch"):
            b = r.read_batch(self.batch)
            m = 0 if b is None else len(b)
            if m:
                host_copy(st[:m], self._check(np.asarray(b)))
        else:
            # the reference's bare read(): one frame at a time (video/embedder.py:19-27).  A reader that fails in the middle
            # of a batch has already delivered the frames before it: they are handed on (in order) and the exception
            # follows them, exactly what the reference's loop would have processed before it died
            m = 0
            try:
                while m < self.batch:
                    f = r.read()
                    if f is None:
                        break
                    np.copyto(st[m], self._check(np.asarray(f)[None])[0])
                    m += 1
            except BaseException as exc:
                self._failed = exc
        if m == 0:
            self.free.put(st)
            return None
        return st[:m], st

    _failed = None

    def run(self):
        try:
            while not self.stop.is_set():
                item = self._next()
                if item is None and self._failed is not None:
                    raise self._failed
                self.ready.put(item)
                if item is None:
                    return
                if self._failed is not None:              # the frames read before the failure went first
                    raise self._failed
        except BaseException as exc:                      # handed to the caller's thread, which re-raises it
            self.ready.put(exc)

    def shutdown(self):
        self.stop.set()
        self.free.put(None)
        try:
            while True:
                self.ready.get_nowait()
        except queue.Empty:
            pass


def _tensor_over(array):
    """torch view of a host array for a DMA copy.  A reader may hand out read-only arrays (np.frombuffer over a pipe's
    bytes): torch warns that writes through the tensor would be undefined -- it is only ever a copy SOURCE here."""
    import warnings
    import torch
    if array.flags.writeable:
        return torch.from_numpy(array)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        return torch.from_numpy(array)


class _Slot:
    def __init__(self, torch, device, batch, in_shape, in_dtype, out_shape, out_dtype):
        self.dev_in = torch.empty((batch,) + tuple(in_shape), dtype=in_dtype, device=device)
        self.dev_out = torch.empty((batch,) + tuple(out_shape), dtype=out_dtype, device=device)
        self.host_out = None                            # page-locked landing buffer, only if the sink reserves nothing
        self.ev_in, self.ev_k, self.ev_out = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self.busy = None                                # (count, staging, landing array, reserved?) while in flight


class StagedPipeline:
    """reader -> [upload stream] -> process() on the kernel stream -> [download stream] -> sink, ``depth`` batches in flight.

    process(dev_in[:m], dev_out[:m]) enqueues the kernels on the current stream.
    sink: object with reserve(m) -> page-locked ndarray | None and commit(m), and/or deliver(array [m, ...]).
    """

    def __init__(self, device, reader, batch, in_shape, out_shape, out_dtype=np.uint8, depth=2):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.reader, self.batch, self.depth = reader, int(batch), int(depth)
        self.in_shape, self.out_shape, self.out_dtype = tuple(in_shape), tuple(out_shape), np.dtype(out_dtype)
        self.frames_done = 0

    def run(self, process, sink):
        t = self.torch
        with t.cuda.device(self.device):
            return self._run(process, sink)

    def _run(self, process, sink):
        t = self.torch
        tdtype = getattr(t, self.out_dtype.name)
        zero_copy_in = bool(getattr(self.reader, "pinned", False)) and hasattr(self.reader, "read_batch")
        staging = [] if zero_copy_in else [pinned_empty((self.batch,) + self.in_shape) for _ in range(self.depth + 1)]
        ahead = _ReadAhead(self.reader, self.batch, self.in_shape, np.uint8, staging)
        slots = [_Slot(t, self.device, self.batch, self.in_shape, t.uint8, self.out_shape, tdtype) for _ in range(self.depth)]
        s_up, s_k, s_down = t.cuda.Stream(self.device), t.cuda.Stream(self.device), t.cuda.Stream(self.device)
        reserve = getattr(sink, "reserve", None)

        def drain(slot):
            m, st, landing, reserved = slot.busy
            slot.ev_out.synchronize()
            if reserved:
                sink.commit(m)
            else:
                sink.deliver(landing)
            if st is not None:
                ahead.free.put(st)
            slot.busy = None
            self.frames_done += m

        ahead.start()
        try:
            k = 0
            while True:
                try:
                    item = ahead.ready.get_nowait()
                except queue.Empty:                             # the reader is behind: hand finished frames on meanwhile
                    oldest = next((slots[(k + j) % self.depth] for j in range(self.depth) if slots[(k + j) % self.depth].busy), None)
                    if oldest is not None:
                        drain(oldest)
                    item = ahead.ready.get()
                if isinstance(item, BaseException):             # the reader failed: what is in flight is complete -- hand it on, then raise
                    for j in range(self.depth):
                        if slots[(k + j) % self.depth].busy:
                            drain(slots[(k + j) % self.depth])
                    raise item
                if item is None:
                    break
                src, st = item
                m = len(src)
                slot = slots[k % self.depth]
                if slot.busy:
                    drain(slot)
                landing = reserve(m) if reserve else None
                reserved = landing is not None
                if not reserved:
                    if slot.host_out is None:
                        slot.host_out = pinned_empty((self.batch,) + self.out_shape, self.out_dtype)
                    landing = slot.host_out[:m]
                with t.cuda.stream(s_up):
                    s_up.wait_event(slot.ev_k)                 # the kernels that last read this device buffer are done
                    slot.dev_in[:m].copy_(_tensor_over(src), non_blocking=True)
                    slot.ev_in.record()
                with t.cuda.stream(s_k):
                    s_k.wait_event(slot.ev_in)
                    s_k.wait_event(slot.ev_out)                # the download that last read dev_out is done
                    process(slot.dev_in[:m], slot.dev_out[:m])
                    slot.ev_k.record()
                with t.cuda.stream(s_down):
                    s_down.wait_event(slot.ev_k)
                    t.from_numpy(landing).copy_(slot.dev_out[:m], non_blocking=True)
                    slot.ev_out.record()
                slot.busy = (m, st, landing, reserved)
                k += 1
            for j in range(self.depth):                         # oldest first: frames leave in order
                slot = slots[(k + j) % self.depth]
                if slot.busy:
                    drain(slot)
        finally:
            ahead.shutdown()
            for s in (s_up, s_k, s_down):
                s.synchronize()
        return self.frames_done


# ---- adapters between the duck-typed reader / writer objects and the pipeline ----------------------------------------
class PeekedReader:
    """A reader that offers no ``height`` / ``width``: its first frame is read to learn them and replayed."""

    def __init__(self, reader):
        self.reader = reader
        self.first = reader.read()
        self.pix_fmt = pix_fmt_of(reader)
        if self.first is not None:
            shape = np.asarray(self.first).shape
            self.height, self.width = (shape[0], shape[1]) if self.pix_fmt == "rgb24" else (shape[0] * 2 // 3, shape[1])

    def read(self):
        if self.first is not None:
            f, self.first = self.first, None
            return f
        return self.reader.read()

    def close(self):
        self.reader.close()


class WriterSink:
    """Frames leaving the pipeline -> a frame writer (reserve/commit when it has them, else write_batch, else write)."""

    def __init__(self, writer):
        self.writer = writer
        if hasattr(writer, "reserve") and hasattr(writer, "commit"):
            self.reserve, self.commit = writer.reserve, writer.commit

    def deliver(self, frames):
        if hasattr(self.writer, "write_batch"):
            self.writer.write_batch(frames)
        else:
            for f in frames:
                self.writer.write(f)


def to_rgb_on_device(engine, batch, pix_fmt, height, width):
    """Device batch as delivered by a reader -> interleaved RGB [m, H, W, 3] (no-op for rgb24)."""
    if pix_fmt == "rgb24":
        return batch
    return engine.yuv420_to_rgb(batch.view(batch.shape[0], -1), height, width, layout=PLANAR_LAYOUT[pix_fmt])
