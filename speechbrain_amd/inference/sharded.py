"""Utterance-sharded offline transcription over the GPUs of one node.

The path has no cross-utterance state at inference, so utterances shard embarrassingly: one
process per GPU, each with a full model replica.  Rank 0 owns the job: it sorts utterances by
duration, forms length-bucketed batches (cf. DynamicBatchSampler, dataio/sampler.py:321),
assigns batches to ranks longest-processing-time-first, SCATTERS the padded waveforms (RCCL
scatter over xGMI when the backend is "nccl"; gloo in the CPU tests) and GATHERS the token ids.
Those two collectives are the only communication; nothing is exchanged while decoding.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def plan_batches(n_samples: Sequence[int], max_utts: int = 32, max_padded_samples: Optional[int] = None) -> List[List[int]]:
    """Duration-sorted batches: consecutive runs of the length-sorted utterances, at most
    ``max_utts`` each and (optionally) at most ``max_padded_samples`` of padded audio."""
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


def pad_batch(wavs: Sequence[torch.Tensor], idx: Sequence[int]):
    """batch_pad_right (utils/data_utils.py:459-519): zero right-padding, relative lengths."""
    n = max(wavs[i].numel() for i in idx)
    out = torch.zeros(len(idx), n, dtype=torch.float32)
    for r, i in enumerate(idx):
        out[r, : wavs[i].numel()] = wavs[i]
    return out, torch.tensor([wavs[i].numel() / n for i in idx], dtype=torch.float32)


class ShardedTranscriber:
    """transcribe(wavs on rank 0) -> token-id lists on rank 0, in the input order.

    ``transcribe_batch(wavs [B,N] on `device`, wav_lens [B]) -> list[list[int]]`` is the per-rank
    worker, e.g. ``lambda w, l: asr.transcribe_batch(w, l)[1]``.
    """

    def __init__(self, transcribe_batch: Callable, device, max_utts: int = 32, max_padded_samples: Optional[int] = None,
                 group=None, concurrent=None):
        """``concurrent`` (optional): a ``speechbrain_amd.inference.streams.ConcurrentTranscriber``; the rank's
        batches then run several at a time on separate HIP streams instead of one after the other."""
        self.fn, self.device, self.max_utts, self.max_padded = transcribe_batch, torch.device(device), max_utts, max_padded_samples
        self.group, self.concurrent = group, concurrent
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    # -- scatter -----------------------------------------------------------------
    def scatter(self, wavs: Optional[Sequence[torch.Tensor]]):
        """Rank 0 passes the waveforms; every rank returns its local list of
        (global utterance ids, padded batch on device, relative lengths)."""
        if self.world == 1:
            n = [w.numel() for w in wavs]
            batches = plan_batches(n, self.max_utts, self.max_padded)
            return [(b,) + tuple(t.to(self.device) for t in pad_batch(wavs, b)) for b in batches]
        meta: List = [None] * self.world
        payloads: List[torch.Tensor] = []
        if self.rank == 0:
            n = [w.numel() for w in wavs]
            batches = plan_batches(n, self.max_utts, self.max_padded)
            owner = assign_batches([batch_cost(n, b) for b in batches], self.world)
            flat = []
            for r in range(self.world):
                m, parts = [], []
                for bi in owner[r]:
                    x, lens = pad_batch(wavs, batches[bi])
                    m.append((batches[bi], tuple(x.shape), lens.tolist()))
                    parts.append(x.reshape(-1))
                meta[r] = m
                flat.append(torch.cat(parts) if parts else torch.zeros(0))
            width = max(1, max(f.numel() for f in flat))
            payloads = [torch.nn.functional.pad(f, (0, width - f.numel())).to(self.device) for f in flat]
            sizes = [width]
        else:
            sizes = [None]
        dist.broadcast_object_list(sizes, src=0, group=self.group)
        my_meta = [None]
        dist.scatter_object_list(my_meta, meta if self.rank == 0 else None, src=0, group=self.group)
        recv = torch.empty(sizes[0], dtype=torch.float32, device=self.device)
        dist.scatter(recv, payloads if self.rank == 0 else None, src=0, group=self.group)
        local, off = [], 0
        for ids, shape, lens in my_meta[0]:
            cnt = shape[0] * shape[1]
            local.append((ids, recv[off: off + cnt].view(shape), torch.tensor(lens, dtype=torch.float32, device=self.device)))
            off += cnt
        return local

    # -- run + gather ------------------------------------------------------------
    def run_local(self, local):
        out = []
        if self.concurrent is not None and len(local) > 1:
            for (ids, _, _), hyps in zip(local, self.concurrent.transcribe_batches([(x, l) for _, x, l in local])):
                out.extend(zip(ids, hyps))
            return out
        for ids, x, lens in local:
            hyps = self.fn(x, lens)
            out.extend(zip(ids, hyps))
        return out

    def gather(self, results, n_total: Optional[int] = None):
        """results: list of (utterance id, token list) -> on rank 0 the hypotheses in input order."""
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
