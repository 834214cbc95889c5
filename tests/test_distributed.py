"""Utterance sharding across ranks (speechbrain_amd/inference/sharded.py) with the gloo backend,
world_size 2, on CPU -- the same pattern the reference uses (tests/unittests/test_distributed.py:10-23).
The per-rank worker is a deterministic stand-in: this test covers partitioning, the scatter of padded
waveforms and the gather of token ids, not the kernels."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speechbrain_amd.inference.sharded import ShardedTranscriber, assign_batches, batch_cost, plan_batches


def fake_transcribe(wavs, lens):
    """tokens = [number of valid samples // 1000, first sample * 1e3 rounded] -- depends on content and length."""
    out = []
    for w, l in zip(wavs, lens):
        n = int(round(float(l) * w.numel()))
        out.append([n // 1000, int(round(float(w[0]) * 1000)) % 997, n % 7])
    return out


def make_job(n=23, seed=5):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(2000, 9000, (n,), generator=g).tolist()
    return [torch.rand(m, generator=g) for m in lens]


def expected(wavs):
    return [[w.numel() // 1000, int(round(float(w[0]) * 1000)) % 997, w.numel() % 7] for w in wavs]


def _worker(rank, world, tmpdir, q):
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), str(rank), str(world)
    dist.init_process_group("gloo", init_method=f"file://{tmpdir}/sync", rank=rank, world_size=world)
    try:
        wavs = make_job() if rank == 0 else None
        st = ShardedTranscriber(fake_transcribe, "cpu", max_utts=4)
        local = st.scatter(wavs)
        n_local = sum(len(t[0]) for t in local)
        hyps = st.gather(st.run_local(local))
        if rank == 0:
            q.put(("hyps", hyps))
        q.put(("count", rank, n_local))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_scatter_gather_world2(tmp_path, world):
    """Streamed scatter (one send per batch, round-robin over the peers) + gather, 2 and 3 ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world + 1)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hyps = [g[1] for g in got if g[0] == "hyps"][0]
    counts = {g[1]: g[2] for g in got if g[0] == "count"}
    assert hyps == expected(make_job())          # every utterance, in input order, on rank 0
    assert sum(counts.values()) == 23 and min(counts.values()) >= 4  # every rank did real work


def _worker_variant(rank, world, tmpdir, q, variant):
    import speechbrain_amd.inference.sharded as sharded

    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), str(rank), str(world)
    dist.init_process_group("gloo", init_method=f"file://{tmpdir}/sync", rank=rank, world_size=world)
    try:
        wavs = make_job() if rank == 0 else None
        if variant == "rank0_idle":  # the plan gives rank 0 NO batch: it only stages and sends (from gather's drain)
            lpt = sharded.assign_batches

            def peers_only(costs, w):
                out = lpt(costs, w - 1)
                return [[]] + out

            sharded.assign_batches = peers_only
        st = ShardedTranscriber(fake_transcribe, "cpu", max_utts=4)
        local = st.scatter(wavs)
        n_local = sum(len(t[0]) for t in local)
        if variant == "last_first" and rank != 0:
            # a consumer that asks for the batch that arrives LAST first (a worker pool is free to): every receive was posted
            # in distribute(), so waiting out of order must neither deadlock nor mix the buffers up
            results = []
            for t in reversed(local):
                t[3]()
                results.extend(zip(t[0], fake_transcribe(t[1], t[2])))
            hyps = st.gather(results)
        elif variant == "last_first":
            # rank 0 consumes its own list directly, without run_local: ready() stages on the calling thread
            results = []
            for t in reversed(local):
                t[3]()
                results.extend(zip(t[0], fake_transcribe(t[1], t[2])))
            hyps = st.gather(results)
        else:
            hyps = st.gather(st.run_local(local))
        if rank == 0:
            q.put(("hyps", hyps))
        q.put(("count", rank, n_local))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["rank0_idle", "last_first"])
def test_scatter_edge_cases_world3(tmp_path, variant):
    """(a) rank 0 owns no local batch -- it pads and sends everyone else's while idle itself; (b) consumers
    take their batches in the reverse of the arrival order, rank 0 directly from distribute()'s list without run_local."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_variant, args=(r, world, str(tmp_path), q, variant)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world + 1)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hyps = [g[1] for g in got if g[0] == "hyps"][0]
    counts = {g[1]: g[2] for g in got if g[0] == "count"}
    assert hyps == expected(make_job())
    assert sum(counts.values()) == 23
    if variant == "rank0_idle":
        assert counts[0] == 0 and min(counts[1], counts[2]) >= 4


def test_single_process_path():
    st = ShardedTranscriber(fake_transcribe, "cpu", max_utts=5)
    wavs = make_job(11, seed=9)
    assert st.transcribe(wavs) == expected(wavs)


def test_planning_properties():
    n = [int(v) for v in torch.randint(80000, 480000, (200,), generator=torch.Generator().manual_seed(1))]
    batches = plan_batches(n, max_utts=32, max_padded_samples=32 * 480000)
    assert sorted(i for b in batches for i in b) == list(range(200))       # a partition
    assert all(len(b) <= 32 for b in batches)
    for b in batches:                                                         # duration-sorted buckets: little padding
        assert max(n[i] for i in b) - min(n[i] for i in b) <= 0.35 * max(n)
    costs = [batch_cost(n, b) for b in batches]
    owner = assign_batches(costs, 8)
    assert sorted(i for o in owner for i in o) == list(range(len(batches)))
    loads = [sum(costs[i] for i in o) for o in owner]
    assert max(loads) <= min(loads) + max(costs)                              # LPT balance bound
    assert plan_batches([], 4) == [] and assign_batches([], 3) == [[], [], []]


def _tiny_asr():
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from emu_utils import attach

    attach()  # CPU kernel emulator (test tooling): the product itself has no CPU path
    from speechbrain_amd.inference.builders import build_asr

    tiny = dict(d_model=32, nhead=4, d_ffn=64, n_enc=2, n_dec=2, n_fft=512, win_length=32)
    asr = build_asr(tiny, vocab=40, seed=3, beam_size=3, ctc_weight=0.4, device="cpu", max_decode_ratio=0.4)
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(6.0)
        asr.mods.ctc_lin.w.weight.mul_(6.0)
    return asr


def _asr_job(pcm=False):
    g = torch.Generator().manual_seed(11)
    wavs = [0.1 * torch.randn(int(n), generator=g) for n in torch.randint(3000, 8000, (9,), generator=g)]
    if pcm:  # what a 16-bit wav file holds
        wavs = [(w * 32768.0).round().clamp(-32768, 32767).to(torch.int16) for w in wavs]
    return wavs


def _worker_asr(rank, world, tmpdir, q, pcm=False):
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = str(rank), str(rank), str(world)
    dist.init_process_group("gloo", init_method=f"file://{tmpdir}/sync", rank=rank, world_size=world)
    try:
        from speechbrain_amd.inference.streams import ConcurrentTranscriber

        asr = _tiny_asr()
        st = ShardedTranscriber(lambda w, l: asr.transcribe_batch(w, l)[1], "cpu", max_utts=2,
                                concurrent=ConcurrentTranscriber(asr, streams=3, group=2 if pcm else 1))
        hyps = st.transcribe(_asr_job(pcm) if rank == 0 else None)
        if rank == 0:
            q.put((hyps, st.last_plan["bytes_sent"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pcm", [False, True])
def test_sharded_asr_world2_matches_single_process(tmp_path, pcm):
    """The whole multi-rank path with the real modules (tiny Conformer on the CPU kernel emulator): rank 0
    scatters padded waveforms, both ranks transcribe their batches through ConcurrentTranscriber, token ids
    are gathered -- and equal a plain single-process transcription of the same batches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_asr, args=(r, 2, str(tmp_path), q, pcm)) for r in range(2)]
    for p in procs:
        p.start()
    hyps, bytes_sent = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from speechbrain_amd.inference.sharded import pad_batch

    asr = _tiny_asr()
    wavs = _asr_job(pcm)
    ref = [None] * len(wavs)
    for b in plan_batches([w.numel() for w in wavs], max_utts=2):
        x, lens = pad_batch(wavs, b)
        if pcm:  # int16 PCM travels as int16 (2 bytes per sample) and becomes sample / 32768 on the receiving rank
            x = x.float() / 32768.0
        for i, h in zip(b, asr.transcribe_batch(x, lens)[1]):
            ref[i] = h
    assert hyps == ref
    sent_samples = bytes_sent // (2 if pcm else 4)
    assert 0 < sent_samples <= sum(w.numel() for w in wavs) * 1.3  # exact-size sends: no padding to the widest rank


def test_bench_self_launch_world2():
    """`python bench.py --gpus 2` with no rendezvous in the environment must start two ranks itself (torch.distributed.run
    on 127.0.0.1), run the sharded path and print ONE line with n_gpus = 2.  --launch-check keeps everything but the GPU
    work: gloo instead of RCCL, a stand-in transcriber, the real plan / streamed scatter / gather."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["utterances_total"] == 256 and rec["ranks_with_work"] == 2
    assert rec["bytes_scattered"] > 0


def test_bench_node_job_world8_launch_check():
    """BASELINE.json configs[3] as far as a box without GPUs can take it: `bench.py --gpus 8 --job-utts 10000 --launch-check`
    starts eight ranks (gloo), plans the 10 000-utterance job (durations scaled to a tenth), streams every batch to its
    owner, runs the stand-in transcriber on every rank and gathers 10 000 hypotheses in input order on rank 0.  The plan
    is the real one: 313 batches, every rank owns some, and the longest-processing-time-first assignment leaves the
    most loaded rank within 1 % of the mean cost."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--job-utts", "10000", "--warmup", "0",
                          "--launch-check"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 8 and rec["utterances_total"] == 10000 and rec["ranks_with_work"] == 8
    assert rec["batches_total"] == 313 and rec["plan_max_over_mean_cost"] <= 1.01
