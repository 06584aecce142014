import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-fingerprinting_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import offmark_oracle as orc
from offmark.engine import DctEngine
eng = DctEngine()
for case in sys.argv[1:]:
    g = np.load(os.path.join(ROOT, "tests/golden", case + ".npz"))
    frame, alpha = g["frame"], float(g["alpha"])
    enc = orc.DctEncoderOracle(alpha=alpha); enc.read_wm(g["wm"])
    enc.encode(orc.bgr2yuv_f32(frame.astype(np.float32)))
    d = eng.debug_planes(torch.from_numpy(frame).cuda(), alpha=alpha, wm=g["wm"])
    o = enc.debug
    print(case, "max|ydc|", np.abs(d["y_dc"] - o["ydc"]).max(), "max|c21|", np.abs(d["c21_pre"] - o["c21_pre"]).max())
    bad = (np.abs(d["step"] - alpha * o["mask"]) > 1e-4) | (np.abs(np.abs(d["c21_post"]) - np.abs(o["c21_post"])) > 2e-3)
    for (i, j) in np.argwhere(bad):
        print(f"  blk {i},{j}: c21 gpu {d['c21_pre'][i,j]:.6g} ora {o['c21_pre'][i,j]:.6g} | step gpu {d['step'][i,j]:.9g} ora {alpha*o['mask'][i,j]:.9g}"
              f" | lum {d['lum'][i,j]:.9g}/{o['lum'][i,j]:.9g} tex {d['tex'][i,j]:.9g}/{o['tex'][i,j]:.9g} | post {d['c21_post'][i,j]:.6g}/{o['c21_post'][i,j]:.6g}")
