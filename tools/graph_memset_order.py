"""Three steps of (ofmk_svd_detect_rgb8 + ofmk_payloads_from_counts) captured into ONE hipGraph and replayed four times, with the
steps sharing one counts buffer ("shared"), using one each ("distinct"), or going through DctEngine (torch's allocator decides).

Why it exists: until round 4 the library zeroed counts / accumulators with hipMemsetAsync.  On ROCm 7.2 a graph holding
memset(c) -> kernels(c) -> memset(c) -> kernels(c) ... replays with the memset nodes out of order from the SECOND replay on
(profiles/r4_graph_memset_order.txt: "shared" right on replay 0, steps 0 and 2 wrong on replays 1-3; "distinct" always right).
bench.py's grouped steps (several steps per graph, one allocator block reused) hit exactly that.  The library now zero-fills with a
kernel of its own (csrc/offmark_kernels.hip: launch_zero): kernel nodes keep their order.  With the current library every line
this prints must be all True (tests/test_gpu_parity.py::test_several_steps_in_one_graph_replay_like_eager is the pinned form)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-fingerprinting_amd"))
import numpy as np, torch
from offmark import _hip
from offmark.engine import DctEngine
from offmark.degenerator.de_shuffler import DeShuffler
from offmark.generator.shuffler import Shuffler
from offmark.synthetic import synthetic_frames
H, W, n, L = 240, 320, 48, 8
N = H * W // 64
P = np.array([0, 1, 1, 0, 0, 1, 0, 1])
eng = DctEngine()
lib = eng.lib
src = synthetic_frames(n, H, W, seed=5)
wm = Shuffler(key=0).generate_wm(P, (1, N)).astype(np.uint8)
frames = eng.svd_embed(src, wm, scale=15)
perm = torch.as_tensor(DeShuffler(key=0).set_shape(P.shape).payload_idx, dtype=torch.int32).cuda()
ref_c, _ = eng.svd_detect(frames, L, scale=15)
ref_p = eng.payloads(ref_c, N, perm)
torch.cuda.synchronize()
print("eager payload ok:", bool((ref_p.cpu().numpy() == P).all()))
sc = _hip.scales3(15)
def step(counts, pay):
    _hip.check(lib.ofmk_svd_detect_rgb8(frames.data_ptr(), n, H, W, L, sc, 4, counts.data_ptr(), None, _hip.current_stream(), None))
    _hip.check(lib.ofmk_payloads_from_counts(counts.data_ptr(), n, L, N, perm.data_ptr(), pay.data_ptr(), _hip.current_stream(), None))
for mode in ("shared", "distinct", "engine"):
    G = 3
    pay = torch.zeros((G, n, L), dtype=torch.uint8, device="cuda")
    cs = [torch.empty((n, L), dtype=torch.int32, device="cuda") for _ in range(G)]
    s = torch.cuda.Stream()
    def body():
        for g in range(G):
            if mode == "engine":
                c, _ = eng.svd_detect(frames, L, scale=15)
                eng.payloads(c, N, perm, out=pay[g])
            else:
                step(cs[0] if mode == "shared" else cs[g], pay[g])
    with torch.cuda.stream(s):
        body(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            body()
    torch.cuda.synchronize()
    for rep in range(4):
        pay.zero_()
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        ok = [bool(torch.equal(pay[g], ref_p)) for g in range(G)]
        print(mode, "replay", rep, ok, "counts equal:", [bool(torch.equal(c, ref_c)) for c in cs] if mode != "engine" else "")
