"""offmark.dist.steps.StepPipeline on the GPU: grouped steps replayed as one hipGraph on two branches must give, step for step,
what the plain one-step-at-a-time loop gives (payloads, votes, marked frames), for groups that divide the step count and for
a ragged last group; the reference shape behind it is the per-segment loop of tests/segment_mark_detect_hls.py:407-412 with
its vote (:126-155)."""
import numpy as np
import pytest

import offmark_oracle as orc

pytestmark = pytest.mark.gpu
P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])


def _pipeline(codec, group, graph, lanes=1):
    import torch
    from offmark.degenerator.de_shuffler import DeShuffler
    from offmark.dist.steps import StepPipeline
    from offmark.engine import DctEngine
    from offmark.synthetic import synthetic_frames
    H, W, n, L = 64, 96, 10, 8
    N = H * W // 64
    frames = synthetic_frames(n, H, W, seed=321)
    seg = np.repeat([0, 1], n // 2)                                   # two segments of five frames, own payload each
    wm = torch.from_numpy(np.stack([orc.shuffle_generate(P8, (N,), 0), orc.shuffle_generate(1 - P8, (N,), 0)]).astype(np.uint8)).cuda()
    rows = torch.from_numpy(seg.astype(np.int32)).cuda()
    perm = torch.as_tensor(DeShuffler(key=0).set_shape((L,)).payload_idx, dtype=torch.int32).cuda()

    def issue(e, out, slot):
        if codec == "dct":
            _, c, _ = e.embed_detect(frames, wm, L=L, wm_row=rows, out=out)
        else:
            _, c, _ = e.svd_embed_detect(frames, wm, L=L, wm_row=rows, out=out)
        e.payloads(c, N, perm, out=slot)
    p = StepPipeline("cuda", n, L, seg, make_engine=DctEngine, make_out=lambda: torch.empty_like(frames), issue=issue,
                     lanes=lanes, group=group, graph=graph)
    p.prepare()
    return p


@pytest.mark.parametrize("codec", ["dct", "dwtdctsvd"])
def test_grouped_graph_steps_equal_the_plain_loop(codec):
    import torch
    plain = _pipeline(codec, 1, False)
    v_plain, size = plain.run(4)
    assert size == 1 and sorted(v_plain) == [0, 1]
    want = {s: v_plain[s][0].tolist() for s in v_plain}
    assert want == {0: P8.tolist(), 1: (1 - P8).tolist()} and all(v_plain[s][1] == 1.0 for s in v_plain)
    ref_pay = plain.last_payloads().clone()
    ref_out = plain.lanes[0].out.clone()
    for group, steps, lanes in ((3, 6, 1), (3, 7, 1), (4, 9, 2), (2, 2, 1)):
        p = _pipeline(codec, group, True, lanes)
        assert p.plan(steps) == [group] * (steps // group) + ([steps % group] if steps % group else [])
        assert all(l.eng2 is not None and l.graph[0] is not None and l.graph[1] is not None for l in p.lanes)     # two branches, both halves captured
        votes, size = p.run(steps)
        assert size == (steps % group or group)
        S = 2
        assert sorted(votes) == [k * S + s for k in range(size) for s in range(S)] if size > 1 else sorted(votes) == [0, 1]
        for key, (pattern, freq) in votes.items():
            assert pattern.tolist() == want[key % S] and freq == 1.0
        lane, par = p.last
        for k in range(size):
            assert torch.equal(lane.pay[par, k], ref_pay)                                   # every step of the group, bit for bit
        assert torch.equal(lane.out, ref_out) and (lane.out2 is None or torch.equal(lane.out2, ref_out))
        assert p.host_s["enqueue"] > 0 and p.host_s["vote"] > 0
        del p
    import gc
    gc.collect()
    torch.cuda.synchronize()
