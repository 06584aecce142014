"""One worker of bench.py's CPU baseline variants A and B2: the NumPy oracle on ONE core,
embed+detect of `frames` synthetic frames -- form "vec" (default; all blocks at once, variant B2) or "loop" (the
reference-shaped per-block Python loops of dct_encoder.py:18-102 / dct_decoder.py:10-27, variant A).  TEST INFRASTRUCTURE ONLY (oracle/): started as a separate
process per host core by bench.py's cpu_baseline leg; never touches the GPU, never imported by the product.

usage: cpu_baseline_worker.py H W frames seed alpha [form]   ->  one JSON line {"t0", "t1", "ok"} (wall-clock span of the
timed part, so the parent can take the union over workers)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import offmark_oracle as orc  # noqa: E402

H, W, n, seed, alpha = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
form = sys.argv[6] if len(sys.argv) > 6 else "vec"
P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
frames = [orc.synthetic_frame(H, W, seed + i) for i in range(n)]          # untimed set-up
wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)
enc = orc.DctEncoderOracle(alpha=alpha, form=form)
enc.read_wm(wm)
ok = True
t0 = time.time()
for f in frames:
    bits = orc.check_frame(orc.mark_frame(f, enc), orc.DctDecoderOracle(alpha=alpha, form=form))
    ok &= bool(np.array_equal(orc.deshuffle(bits, 8, 0), P8))
t1 = time.time()
print(json.dumps(dict(t0=t0, t1=t1, ok=ok)))
