"""World-size-2 test of the multi-GPU path on CPU (gloo): frames shard contiguously, each rank
recovers its own frames' payloads, one all-gather, the same vote on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import offmark_oracle as orc  # noqa: F401  (conftest puts oracle/ and the package on sys.path)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, ragged, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "video-fingerprinting_amd")]
    from offmark.dist.vote import gather_payloads, shard_range, vote_segments
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)                       # same stream on every rank
        segments = np.repeat(np.arange(4), n_total // 4 + 1)[:n_total]
        truth = np.array([[int(b) for b in format(s + 1, "08b")] for s in segments], dtype=np.uint8)
        noisy = truth.copy()
        flip = rng.random(n_total) < 0.2                      # a fifth of the frames decode wrongly
        noisy[flip] ^= rng.integers(0, 2, size=(int(flip.sum()), 8)).astype(np.uint8)
        a, b = shard_range(n_total, rank, world) if ragged else (rank * (n_total // world), (rank + 1) * (n_total // world))
        mine = torch.from_numpy(noisy[a:b])
        everyone = gather_payloads(mine).numpy()
        used = n_total if ragged else (n_total // world) * world
        assert everyone.shape == (used, 8) and np.array_equal(everyone, noisy[:used])
        votes = vote_segments(everyone, segments[:used])
        q.put((rank, {k: (v[0].tolist(), v[1]) for k, v in votes.items()}))
    finally:
        dist.destroy_process_group()


def _run(n_total, ragged, world=2):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(out[r] == out[0] for r in range(world))        # identical vote on every rank
    for seg, (pattern, freq) in out[0].items():
        assert pattern == [int(b) for b in format(seg + 1, "08b")] and freq >= 0.5
    return out


def test_equal_shards_all_gather_and_vote():
    _run(48, ragged=False)


def test_ragged_shards_all_gather_and_vote():
    _run(37, ragged=True)


def test_four_ranks_ragged():
    _run(50, ragged=True, world=4)


def test_eight_ranks_one_segment_each_config4_shape():
    """BASELINE config 4 at its full width: 8 segments x 48 frames, segment s on rank s (equal shards), payload
    format(s + 1, '08b'); every rank must end with every segment's payload (tests/segment_mark_detect_hls.py:126-155)."""
    n_total = 8 * 48
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, n_total, q)) for r in range(8)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=240) for _ in range(8))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(out[r] == out[0] for r in range(8))
    assert sorted(out[0]) == list(range(8))
    for seg, (pattern, freq) in out[0].items():
        assert pattern == [int(b) for b in format(seg + 1, "08b")] and freq >= 0.5


def _worker8(rank, world, port, n_total, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "video-fingerprinting_amd")]
    from offmark.dist.vote import gather_payloads, shard_range, vote_segments
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per = n_total // world
        assert shard_range(n_total, rank, world) == (rank * per, (rank + 1) * per)     # segment s -> rank s
        rng = np.random.default_rng(100 + rank)               # each rank only ever sees its own segment
        mine = np.tile(np.array([int(b) for b in format(rank + 1, "08b")], dtype=np.uint8), (per, 1))
        flip = rng.random(per) < 0.25
        mine[flip] ^= rng.integers(0, 2, size=(int(flip.sum()), 8)).astype(np.uint8)
        everyone = gather_payloads(torch.from_numpy(mine), equal_shards=True).numpy()
        assert everyone.shape == (n_total, 8) and np.array_equal(everyone[rank * per:(rank + 1) * per], mine)
        votes = vote_segments(everyone, np.repeat(np.arange(world), per))
        q.put((rank, {int(k): (v[0].tolist(), v[1]) for k, v in votes.items()}))
    finally:
        dist.destroy_process_group()


def _worker_groups(rank, world, port, steps, per, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.join(os.path.dirname(here), "video-fingerprinting_amd")]
    from offmark.dist.vote import gather_payloads, group_segment_ids, vote_groups
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # config 4's shape: one segment per rank; step g marks segment s with payload format((7 * g + s) % 255 + 1): every step of the
        # group carries DIFFERENT payloads, so a vote that mixed steps (or ranks) would be caught
        rng = np.random.default_rng(500 + rank)
        mine = np.zeros((steps, per, 8), dtype=np.uint8)
        for g in range(steps):
            mine[g] = [int(b) for b in format((7 * g + rank) % 255 + 1, "08b")]
            flip = rng.random(per) < 0.2
            mine[g][flip] ^= rng.integers(0, 2, size=(int(flip.sum()), 8)).astype(np.uint8)
        everyone = gather_payloads(torch.from_numpy(mine.reshape(steps * per, 8)), equal_shards=True).numpy()     # ONE collective per group
        assert everyone.shape == (world * steps * per, 8)
        assert np.array_equal(everyone.reshape(world, steps, per, 8)[rank], mine)                                # rank-major
        seg_one_step = np.repeat(np.arange(world), per)
        ids = group_segment_ids(seg_one_step, world, steps)
        assert ids.shape == (world * steps * per,) and ids.reshape(world, steps, per)[rank, 2, 0] == 2 * world + rank
        votes = vote_groups(everyone, seg_one_step, world, steps)
        q.put((rank, [{int(k): (v[0].tolist(), float(v[1])) for k, v in step.items()} for step in votes]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_grouped_steps_one_gather_one_vote_per_group(world):
    """Small shards are issued several steps per host iteration (bench.py: a 48-frame segment is 0.2 ms of GPU work, less
    than the host needs per step): each rank contributes steps x n payload rows, ONE all-gather returns them rank-major, ONE
    vote_segments call resolves every (step, segment).  Every rank must end with every step's every segment's own payload."""
    steps, per = 5, 12
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_groups, args=(r, world, port, steps, per, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(out[r] == out[0] for r in range(world))
    assert len(out[0]) == steps
    for g, step in enumerate(out[0]):
        assert sorted(step) == list(range(world))
        for seg, (pattern, freq) in step.items():
            assert pattern == [int(b) for b in format((7 * g + seg) % 255 + 1, "08b")] and freq >= 0.5


@pytest.mark.skipif("__import__('torch').cuda.is_available()")
def test_bench_starts_its_own_ranks_and_relays_their_failure():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (a child torch.distributed.run;
    the parent never touches the GPU).  Without a GPU the ranks must fail loudly (no CPU fallback), and the parent must
    hand that on: non-zero exit code, no JSON line.  The working two-rank run is the -m gpu test
    test_bench_two_ranks_gloo_on_one_device[...-self]."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                        "--height", "240", "--width", "320", "--frames", "8", "--steps", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "local_rank: 1" in r.stderr or "rank      : 1" in r.stderr          # two ranks were really started


def test_step_pipeline_refuses_grouped_steps_over_ragged_shards():
    """with group > 1 the gathered layout is [rank][step][n] with ONE n for every rank; ragged shards (8 segments over
    3 ranks: 3 + 3 + 2) or an empty shard would mix steps and segments in the vote without any error.  The constructor says so
    before it touches a device (this runs without a GPU)."""
    from offmark.dist.steps import StepPipeline
    seg = np.repeat(np.arange(8), 4)                       # 8 segments x 4 frames over 3 ranks: this rank holds 3 segments = 12 rows
    for kwargs in (dict(n=12, equal_shards=False), dict(n=0, equal_shards=True), dict(n=5, equal_shards=True)):
        with pytest.raises(ValueError, match="equal, non-empty shards"):
            StepPipeline("cpu", L=8, segment_ids=seg, make_engine=None, make_out=None, issue=None, group=4, **kwargs)


def test_bench_placement_helpers_without_a_gpu(tmp_path, monkeypatch):
    """bench.py binds every rank to its GPU's NUMA-local cores before anything touches the GPU.  Host logic
    only: the cpulist parser, the split of one node's cores between the ranks that share it, and that a host which exposes no
    topology leaves the process alone instead of failing the run."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    assert bench._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and bench._cpulist("") == set()
    before = os.sched_getaffinity(0)
    try:
        cores = sorted(before)
        monkeypatch.setattr(bench, "gpu_numa_nodes", lambda: [0, 0, -1])
        real_open = open

        def fake_open(path, *a, **k):
            if str(path) == "/sys/devices/system/node/node0/cpulist":
                import io
                return io.StringIO(",".join(map(str, cores)))
            return real_open(path, *a, **k)
        monkeypatch.setattr("builtins.open", fake_open)
        monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
        monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
        got, masks = [], []
        for r in (0, 1):                                    # (each rank is its own process in a real run: undo the first binding)
            os.sched_setaffinity(0, before)
            got.append(bench.bind_to_gpu_numa([0, 1, 2], r))
            masks.append(os.sched_getaffinity(0))
        if len(cores) >= 4:                                 # two ranks on node 0: disjoint halves of its cores
            half = len(cores) // 2
            assert got[0]["bound"] and got[1]["bound"] and got[0]["ranks_on_node"] == 2 and got[0]["n_cpus"] == half
            assert masks == [set(cores[:half]), set(cores[half:2 * half])]
        os.sched_setaffinity(0, before)
        none = bench.bind_to_gpu_numa([0, 1, 2], 2)         # a GPU that reports no node
        assert not none["bound"] and "no NUMA node" in none["note"] and os.sched_getaffinity(0) == before
        # device i of the process = entry i of HIP_VISIBLE_DEVICES (applied on top of ROCR_VISIBLE_DEVICES): rank 0 -> physical GPU 2 (no node)
        monkeypatch.setattr(bench, "gpu_numa_nodes", lambda: [0, 0, -1, 0])
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
        remapped = [bench.bind_to_gpu_numa([0, 1], r) for r in (0, 1)]
        assert remapped[0]["gpu"] == 2 and not remapped[0]["bound"] and remapped[1]["gpu"] == 0 and remapped[1]["bound"]
        os.sched_setaffinity(0, before)
        monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3,1,0")                  # HIP's list indexes ROCR's: "2,0" -> physical 0 and 3
        both = bench.bind_to_gpu_numa([0, 1], 1)
        assert both["gpu"] == 3 and both["bound"] and both["ranks_on_node"] == 2
        os.sched_setaffinity(0, before)
        monkeypatch.delenv("HIP_VISIBLE_DEVICES")
        monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
        monkeypatch.setattr(bench, "gpu_numa_nodes", lambda: [])
        assert not bench.bind_to_gpu_numa([0], 0)["bound"]   # no topology at all: an error text, no exception
    finally:
        os.sched_setaffinity(0, before)
    # the failure-injection hook of the N > 1 tests
    monkeypatch.setenv("OFMK_BENCH_INJECT_FAILURE", "1:second_pass")
    bench.inject_failure("second_pass", 0)
    with pytest.raises(RuntimeError, match="rank 1 at second_pass"):
        bench.inject_failure("second_pass", 1)


def _load_bench():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_env_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    return bench


def test_both_launch_paths_give_the_ranks_the_same_environment(monkeypatch):
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, which RCCL's cross-process buffer sharing needs on this
    pool's host driver) was set only for ranks bench.py started itself.  Now both paths go through bench.rank_environment:
    the child launcher's environment (self-launch) and every rank's own os.environ before `import torch` (any launcher)."""
    import inspect
    bench = _load_bench()
    assert bench.RANK_ENV == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    env = {}
    assert bench.rank_environment(env, 1) == {"HSA_ENABLE_IPC_MODE_LEGACY": None} and env == {}          # one rank: nothing to share
    assert bench.rank_environment(env, 2) == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"} and env == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    mine = {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}                                                         # an explicit choice of the caller stands
    assert bench.rank_environment(mine, 8) == {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}
    # path 1: bench.py starts its own ranks -- the environment handed to the child torch.distributed.run
    seen = {}

    class FakeChild:
        stdout = []

        def wait(self):
            return 0

    def fake_popen(cmd, env=None, **kw):
        seen.update(cmd=cmd, env=env)
        return FakeChild()
    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    from types import SimpleNamespace
    with pytest.raises(SystemExit):
        bench.launch_ranks(SimpleNamespace(gpus=2))
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["OFMK_BENCH_SELF_LAUNCHED"] == "1"
    assert "torch.distributed.run" in seen["cmd"] and "127.0.0.1" in seen["cmd"]
    # path 2: ranks started by an outside launcher -- main() applies the same function to os.environ BEFORE torch is imported
    src = inspect.getsource(bench.main)
    assert 0 < src.index("rank_environment(os.environ, env_world)") < src.index("import torch")
    assert src.index("launch_ranks(a)") < src.index("rank_environment(os.environ, env_world)")


def test_bench_oracle_check_bookkeeping_and_budgets():
    """bench.py's `oracle_check` (dct_decoder.py:10-27): host logic only -- with the C oracle standing in for the
    GPU on both sides the check must come out clean, and a marked frame that is off by more than one LSB inside a sign-determined
    block, a flipped raw bit beyond the budget or a wrong payload must each turn `within_budget` false."""
    import c_oracle
    import offmark_oracle as orc
    bench = _load_bench()
    P8 = np.array([0, 1, 1, 0, 0, 1, 0, 1])
    H, W = 64, 96
    frames = np.stack([orc.synthetic_frame(H, W, 40 + i) for i in range(3)])
    wm = orc.shuffle_generate(P8, (1, H * W // 64), 0)[0].astype(np.uint8)
    marked = c_oracle.mark_frames(frames, wm, alpha=20, legacy=True, threads=1)[0]
    bits_of = lambda ref: c_oracle.check_frames(ref, alpha=20, legacy=True, threads=1)[0]      # noqa: E731
    pay = np.stack([P8] * 3)
    ok = bench.oracle_check(frames, marked, bits_of, [wm] * 3, 20.0, pay)
    assert ok["within_budget"] and ok["payload_equal"] and ok["frames"] == 3 and ok["raw_bits_differing"] == 0
    assert ok["pixels_differing_over_determined_blocks"] == 0 and ok["pixels_compared"] > 0 and ok["blocks"] == 3 * 96
    # a marked sample three levels off inside a sign-determined block
    enc = orc.DctEncoderOracle(alpha=20)
    enc.read_wm(wm[None])
    enc.encode(orc.bgr2yuv_f32(frames[0].astype(np.float32)))
    bi, bj = np.argwhere(np.abs(enc.debug["c21_pre"]) > 1e-3)[0]
    bad = marked.copy()
    bad[0, bi * 8, bj * 8, 1] = np.clip(int(bad[0, bi * 8, bj * 8, 1]) + 3, 0, 255) if bad[0, bi * 8, bj * 8, 1] < 250 else bad[0, bi * 8, bj * 8, 1] - 3
    off = bench.oracle_check(frames, bad, bits_of, [wm] * 3, 20.0, pay)
    assert not off["within_budget"] and off["max_pixel_difference"] == 3 and off["pixels_differing_over_determined_blocks"] == 1
    # raw bits beyond the budget (floor: one block), and a wrong payload
    flipped = lambda ref: 1 - bits_of(ref)                                                     # noqa: E731
    assert not bench.oracle_check(frames, marked, flipped, [wm] * 3, 20.0, pay)["within_budget"]
    assert not bench.oracle_check(frames, marked, bits_of, [wm] * 3, 20.0, 1 - pay)["within_budget"]
