"""Frames/s of the plugin-level pipeline (Embedder / Extractor over in-memory 1080p frames on the HOST)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark.degenerator.de_shuffler import DeShuffler
from offmark.embed.dct_encoder import DctEncoder
from offmark.extract.dct_decoder import DctDecoder
from offmark.generator.shuffler import Shuffler
from offmark.synthetic import synthetic_frames
from offmark.video.embedder import Embedder
from offmark.video.extractor import Extractor
from offmark.video.frame_reader import ArrayFrameReader
from offmark.video.frame_writer import ArrayFrameWriter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
frames = synthetic_frames(50, 1080, 1920, seed=3).cpu().numpy()
frames = np.concatenate([frames] * (n // 50))
P = np.array([0, 1, 1, 0, 0, 1, 0, 1])
for rep in range(2):
    enc = DctEncoder(); enc.read_wm(Shuffler(key=0).generate_wm(P, enc.wm_capacity((1080, 1920, 3))))
    w = ArrayFrameWriter()
    t0 = time.perf_counter(); Embedder(ArrayFrameReader(frames), enc, w).start(); t1 = time.perf_counter()
    ex = Extractor(ArrayFrameReader(w.frames), DctDecoder(), DeShuffler(key=0).set_shape(P.shape))
    t2 = time.perf_counter(); ex.start(); t3 = time.perf_counter()
    ok = all(np.array_equal(p, P) for p in ex.patterns)
    print(f"rep {rep}: Embedder {len(frames) / (t1 - t0):.0f} frames/s, Extractor {len(frames) / (t3 - t2):.0f} frames/s, payloads ok: {ok}")
