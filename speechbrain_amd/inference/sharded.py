"""Utterance-sharded offline transcription over the GPUs of one node.

The path has no cross-utterance state at inference, so utterances shard embarrassingly: one
process per GPU, each with a full model replica.  Rank 0 owns the job: it sorts utterances by
duration, forms length-bucketed batches (cf. DynamicBatchSampler, dataio/sampler.py:321),
assigns batches to ranks longest-processing-time-first, SCATTERS the padded waveforms (streamed:
one exact-size point-to-point send per batch, round-robin over the peers = over the xGMI links,
RCCL when the backend is "nccl"; gloo in the CPU tests -- a rank starts on its first batch while
the rest is still on its way) and GATHERS the token ids.  Those two exchanges are the only
communication; nothing is exchanged while decoding.

Waveforms travel in the dtype they are given in: int16 PCM (what a wav file holds; 2 bytes per
sample on PCIe and xGMI, converted with sbk_pcm16_to_f32 on the receiving GPU -- the reference's
``sample / 32768`` load convention, dataio/audio_io.py:141-209) or float32.  Nothing is
quantised on the way.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def plan_batches(n_samples: Sequence[int], max_utts: int = 32, max_padded_samples: Optional[int] = None) -> List[List[int]]:
    """Duration-sorted batches: consecutive runs of the length-sorted utterances, at most
    ``max_utts`` each and (optionally) at most ``max_padded_samples`` of padded audio
    (``max_batch_length`` of the reference's DynamicBatchSampler, in samples)."""
    order = sorted(range(len(n_samples)), key=lambda i: (n_samples[i], i))
    batches, cur = [], []
    for i in order:
        longest = n_samples[i]  # sorted ascending: the newcomer is the longest
        if cur and (len(cur) >= max_utts or (max_padded_samples and (len(cur) + 1) * longest > max_padded_samples)):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def batch_cost(n_samples: Sequence[int], batch: Sequence[int], sample_rate: int = 16000, tokens_per_second: float = 4.0):
    """Decode dominates: cost ~ padded seconds * (a + b * decode steps)."""
    sec = max(n_samples[i] for i in batch) / sample_rate
    return len(batch) * sec * (1.0 + 0.25 * tokens_per_second * sec)


def assign_batches(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first: returns, per rank, the batch indices it runs."""
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for b in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(b)
        load[r] += costs[b]
    return out


def pad_batch(wavs: Sequence[torch.Tensor], idx: Sequence[int], out: Optional[torch.Tensor] = None, tails_only: bool = False):
    """batch_pad_right (utils/data_utils.py:459-519): zero right-padding, relative lengths.  Keeps the
    waveforms' dtype (float32 or int16 PCM); ``out``: optional preallocated [len(idx) * n_max] slab."""
    n = max(wavs[i].numel() for i in idx)
    dtype = wavs[idx[0]].dtype
    if out is None:
        out = torch.zeros(len(idx), n, dtype=dtype)
    else:
        out = out.view(len(idx), n)
        if not tails_only:
            out.zero_()
    for r, i in enumerate(idx):
        m = wavs[i].numel()
        out[r, :m] = wavs[i]
        if tails_only and m < n:
            out[r, m:] = 0  # (every byte of the slab is written exactly once)
    return out, torch.tensor([wavs[i].numel() / n for i in idx], dtype=torch.float32)


class ShardedTranscriber:
    """transcribe(wavs on rank 0) -> token-id lists on rank 0, in the input order.

    ``transcribe_batch(wavs [B,N] on `device`, wav_lens [B]) -> list[list[int]]`` is the per-rank
    worker, e.g. ``lambda w, l: asr.transcribe_batch(w, l)[1]``.
    """

    def __init__(self, transcribe_batch: Callable, device, max_utts: int = 32, max_padded_samples: Optional[int] = None,
                 group=None, concurrent=None, prepare: Optional[Callable] = None):
        """``concurrent`` (optional): a ``speechbrain_amd.inference.streams.ConcurrentTranscriber``; the rank's
        batches then run several at a time on separate HIP streams instead of one after the other (``prepare`` is
        handed to it)."""
        self.fn, self.device, self.max_utts, self.max_padded = transcribe_batch, torch.device(device), max_utts, max_padded_samples
        self.group, self.concurrent, self.prepare = group, concurrent, prepare
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._copy_stream = None
        self.last_plan = None  # rank 0: {"batches": ..., "owner": ..., "bytes_sent": ...} of the last scatter
        self._staging = None   # rank 0: the batches distribute() has announced and _stage_all() has not yet produced

    # -- scatter -----------------------------------------------------------------
    def plan(self, wavs: Optional[Sequence[torch.Tensor]]):
        """Rank 0, metadata only (milliseconds): sort by duration, bucket, assign longest-processing-time-first, fix
        the order every rank runs its batches in.  The padding of the batches into pinned host memory is NOT done here:
        it is streamed batch by batch behind the first batches' compute (``distribute`` / ``run_local``).  Other ranks
        pass None and get None."""
        if self.rank != 0:
            return None
        n = [w.numel() for w in wavs]
        batches = plan_batches(n, self.max_utts, self.max_padded)
        owner = assign_batches([batch_cost(n, b) for b in batches], self.world)
        dtype = wavs[0].dtype if len(wavs) else torch.float32
        longest = [max(n[i] for i in b) for b in batches]
        metas = []
        for r in range(self.world):
            owner[r].sort(key=lambda b: (-len(batches[b]) * longest[b], b))  # the order the rank's workers take them in (largest first)
            meta = [(batches[b], (len(batches[b]), longest[b]), [n[i] / longest[b] for i in batches[b]]) for b in owner[r]]
            metas.append((str(dtype), sum(m[1][0] * m[1][1] for m in meta), meta))
        itemsize = torch.empty(0, dtype=dtype).element_size()
        self.last_plan = {"batches": batches, "owner": owner, "bytes_sent": sum(m[1] for m in metas[1:]) * itemsize}
        return {"wavs": wavs, "metas": metas, "dtype": dtype}

    def _pinned_slot(self, k, count, dtype):
        """Slot k of the ring of pinned staging buffers (allocated once per transcriber: pinning memory is a driver
        round trip per call); waits until the copy that last read the slot has finished."""
        ring = getattr(self, "_ring", None)
        if ring is None or ring["dtype"] != dtype or ring["count"] < count:
            for ev in (ring["events"] if ring else []):
                if ev is not None:
                    ev.synchronize()
            ring = {"dtype": dtype, "count": count, "events": [None] * 8,
                    "bufs": [torch.empty(count, dtype=dtype, pin_memory=True) for _ in range(8)]}
            self._ring = ring
        k %= len(ring["bufs"])
        if ring["events"][k] is not None:
            ring["events"][k].synchronize()
        return k, ring["bufs"][k][:count]

    def _stage_all(self):
        """Rank 0: pad -> (pinned ring ->) device -> (peers: exact-size point-to-point send), batch by batch, round-robin
        over the ranks in the order the ranks run their batches.  Runs on the calling thread (every process-group call
        of a rank stays on one thread) WHILE this rank's workers already transcribe the batches that have landed."""
        todo, self._staging = getattr(self, "_staging", None), None
        if not todo:
            return
        try:
            self._stage(todo)
        except BaseException as e:  # a worker blocked in ready() must not wait for a batch that will never come
            for _, _, _, own in todo["ops"]:
                if own is not None and not own["landed"].is_set():
                    own["error"] = e
                    own["landed"].set()
            raise

    def _stage(self, todo):
        cuda = self.device.type == "cuda"
        wavs, copy_stream = todo["wavs"], todo["copy_stream"]
        for k, (r, ids, shape, own) in enumerate(todo["ops"]):
            cnt = shape[0] * shape[1]
            if cuda:
                slot, host = self._pinned_slot(k, todo["max_count"], todo["dtype"])
                pad_batch(wavs, ids, out=host[:cnt], tails_only=True)
                with torch.cuda.stream(copy_stream):
                    if r == 0:
                        own["buf"].view(-1).copy_(host[:cnt], non_blocking=True)
                    else:
                        chunk = host[:cnt].to(self.device, non_blocking=True)
                        self._pending.append((dist.isend(chunk, r, group=self.group), chunk))
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                self._ring["events"][slot] = ev
                if r == 0:
                    own["ev"] = ev
            else:
                x, _ = pad_batch(wavs, ids)
                if r == 0:
                    own["buf"].copy_(x)
                else:
                    self._pending.append((dist.isend(x, r, group=self.group), x))
            if r == 0:
                own["landed"].set()

    def distribute(self, plan):
        """The scatter proper, streamed batch by batch.  Every rank returns its local list of (global utterance ids,
        padded batch on device [B,N] in the input dtype, relative lengths [B], ready) where ``ready()`` -- called
        on the stream (and host thread) that is about to read the batch -- orders that stream after the batch's
        arrival.  On rank 0 nothing has been staged when this returns: ``run_local`` starts the workers and THEN pads /
        copies / sends the batches one by one (``_stage_all``: round-robin over the ranks in the order the ranks will
        run them; a peer's batch is an exact-size point-to-point send over that peer's xGMI link -- RCCL; gloo in the CPU
        tests), so nobody waits for the whole job: a rank starts transcribing when its first batch has landed, and the
        host's padding and rank 0's PCIe traffic overlap everyone's compute.  The per-rank metadata (pickled, small)
        travels first."""
        import threading

        cuda = self.device.type == "cuda"
        if self.world == 1:
            dtype_name, count, my_meta = plan["metas"][0]
        else:
            mine = [None]
            dist.scatter_object_list(mine, plan["metas"] if self.rank == 0 else None, src=0, group=self.group)
            dtype_name, count, my_meta = mine[0]
        dtype = {"torch.int16": torch.int16, "torch.float32": torch.float32}[dtype_name]
        self._pending = []  # (work, tensor) of sends in flight: kept until gather()

        def waiter(work):
            return work.wait  # NCCL: the calling thread's current stream waits; gloo: the host waits

        local = []
        if self.rank == 0:
            if cuda and getattr(self, "_copy_stream", None) is None:
                self._copy_stream = torch.cuda.Stream(self.device)  # ONE copy stream per transcriber (not per call)
            copy_stream = self._copy_stream if cuda else None
            if cuda:  # (the batch buffers below come from the current stream's pool)
                copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            metas = [m[2] for m in plan["metas"]]
            order = list(range(1, self.world)) + [0]  # the peers' batches go out before rank 0 stages its own
            ops = []
            for j in range(max([len(m) for m in metas] + [0])):
                for r in order:
                    if j >= len(metas[r]):
                        continue
                    ids, shape, lens = metas[r][j]
                    own = None
                    if r == 0:
                        buf = torch.empty(shape, dtype=dtype, device=self.device)
                        if cuda:
                            buf.record_stream(copy_stream)
                        own = {"buf": buf, "ev": None, "landed": threading.Event()}

                        def ready(o=own, staging_thread=threading.get_ident()):
                            # the batch is staged by _stage_all on another thread of control: wait for the copy to have
                            # been ENQUEUED (host), then order the consumer's stream after it and record the stream as
                            # a user of the block (the caching allocator must not hand it out while kernels read it).
                            # A caller that consumes the local list itself, without run_local, calls ready() on the very
                            # thread that would have staged: stage now instead of waiting for nobody
                            if not o["landed"].is_set() and threading.get_ident() == staging_thread and getattr(self, "_staging", None):
                                self._stage_all()
                            o["landed"].wait()
                            if o.get("error") is not None:
                                raise RuntimeError("the batch was never staged") from o["error"]
                            if o["ev"] is not None:
                                cur = torch.cuda.current_stream()
                                cur.wait_event(o["ev"])
                                o["buf"].record_stream(cur)

                        local.append((ids, buf, torch.tensor(lens, dtype=torch.float32, device=self.device), ready))
                    ops.append((r, ids, shape, own))
            self._staging = {"ops": ops, "wavs": plan["wavs"], "dtype": dtype, "copy_stream": copy_stream,
                             "max_count": max([o[2][0] * o[2][1] for o in ops] + [1])}
        else:
            for ids, shape, lens in my_meta:
                buf = torch.empty(shape, dtype=dtype, device=self.device)
                work = dist.irecv(buf, 0, group=self.group)
                local.append((ids, buf, torch.tensor(lens, dtype=torch.float32, device=self.device), waiter(work)))
        return local

    def scatter(self, wavs: Optional[Sequence[torch.Tensor]]):
        """Rank 0 passes the waveforms (1-D float32 or int16 PCM tensors): plan + distribute."""
        return self.distribute(self.plan(wavs))

    # -- run + gather ------------------------------------------------------------
    def _floats(self, x):
        if x.dtype == torch.int16:
            from speechbrain_amd import native

            return native.pcm16_to_f32(x)
        return x

    def run_local(self, local):
        out = []
        if self.concurrent is not None and len(local) > 1:
            # the workers start on whatever has landed; THIS thread then stages the rest of the job (rank 0) behind them
            running = self.concurrent.start([(t[1], t[2]) for t in local], prepare=self.prepare,
                                            ready=[t[3] if len(t) > 3 else None for t in local])
            try:
                self._stage_all()
            finally:
                hyps_per_batch = self.concurrent.finish(running)
            for t, hyps in zip(local, hyps_per_batch):
                out.extend(zip(t[0], hyps))
            return out
        self._stage_all()
        for t in local:
            ids, x, lens = t[0], t[1], t[2]
            if len(t) > 3 and t[3] is not None:
                t[3]()
            hyps = self.fn(self._floats(x), lens)
            out.extend(zip(ids, hyps))
        return out

    def _drain_sends(self):
        self._stage_all()  # (a rank 0 without local batches: nothing else has staged the peers' yet)
        for work, _ in getattr(self, "_pending", []):
            work.wait()
        self._pending = []

    def gather(self, results, n_total: Optional[int] = None):
        """results: list of (utterance id, token list) -> on rank 0 the hypotheses in input order."""
        self._drain_sends()
        if self.world == 1:
            table = dict(results)
            return [table[i] for i in range(len(table))]
        width = max([len(h) for _, h in results] + [1])
        cnt = torch.tensor([len(results), width], dtype=torch.int64, device=self.device)
        all_cnt = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(all_cnt, cnt, group=self.group)
        rows = int(max(c[0] for c in all_cnt))
        W = int(max(c[1] for c in all_cnt))
        host = torch.full((max(rows, 1), W + 2), -1, dtype=torch.int32)
        for r, (i, h) in enumerate(results):
            host[r, 0], host[r, 1] = i, len(h)
            if h:
                host[r, 2: 2 + len(h)] = torch.tensor(h, dtype=torch.int32)
        buf = host.to(self.device)  # one copy to the device before the collective
        gathered = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(buf, gathered, dst=0, group=self.group)
        if self.rank != 0:
            return None
        table = {}
        for g in gathered:
            for row in g.cpu().tolist():
                if row[0] >= 0:
                    table[row[0]] = row[2: 2 + row[1]]
        return [table[i] for i in range(len(table))]

    def transcribe(self, wavs: Optional[Sequence[torch.Tensor]]):
        return self.gather(self.run_local(self.scatter(wavs)))
