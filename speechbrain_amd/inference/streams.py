"""Several utterance batches in flight on one MI355X.

The search half of ``EncoderDecoderASR.transcribe_batch`` is a chain of ~65 short, dependent kernels
per decoding step that cannot fill 256 CUs, while the encoder half is a few large GEMMs.  Running a
handful of independent batches concurrently -- one host thread per batch in flight (the C-ABI calls
release the GIL), the encoder on a normal-priority HIP stream and the search on a HIGH-priority one --
lets the encoder GEMMs of one batch fill the gaps of another batch's search without queueing in front
of its latency-critical kernels.  No state is shared between batches (utterances are
independent), so the results are identical to sequential ``transcribe_batch`` calls.
"""
import copy
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence, Tuple

import torch


class ConcurrentTranscriber:
    """``transcribe_batches([(wavs, wav_lens), ...]) -> [token lists per batch]`` with up to ``streams``
    batches in flight.  ``prepare(searcher, wavs)`` (optional) may adjust the per-worker copy of the
    searcher before a batch (e.g. its decode-length ratios); the copies share the model weights."""

    def __init__(self, asr, streams: int = 8, prioritise_search: bool = True, group: int = 1):
        """``group`` > 1: a worker takes up to that many batches (neighbours in the duration order), encodes them one
        after the other and decodes them in ONE grouped search (``S2STransformerBeamSearcher.forward_group``): every
        batch keeps its own padding and step limits, the decoder step sees the rows of all of them."""
        self.asr, self.n, self.device = asr, max(1, int(streams)), asr.device
        self.group = max(1, int(group))
        # True: a group's batches also share one encoder pass (EncoderDecoderASR.encode_group); an int: that many batches per pass.
        # Round 2 (profiles/r02_group_encoder_ab.jsonl, the fp32-MFMA contractions of the time): 1 % slower.  Round 6, with the
        # split-operand contractions (profiles/r06_r_*, r06_s_*): +1.5 % at 12 steps (12 379 / 12 376 / 12 528 / 12 291 against 12 199 /
        # 12 148 / 12 255 / 12 169), +2.7 % at the driver's 20 steps (12 986 against 12 648; two batches per pass: 12 744) -- and 16-19 GB
        # more reserved memory (the pools of eight encoder streams sized for 4 x the activations): on.
        self.group_encoder = True
        if self.device.type != "cuda":
            self.n = 1
        self.searchers = [copy.copy(asr.mods.decoder) for _ in range(self.n)]
        for s in self.searchers:  # several batches already share the GPU: keep each search on one stream
            if hasattr(s, "overlap_ctc"):
                s.overlap_ctc = 3 if self.n == 1 else 0
        if self.device.type == "cuda":
            self.enc_streams = [torch.cuda.Stream(self.device) for _ in range(self.n)]
            self.dec_streams = [torch.cuda.Stream(self.device, priority=-1) if (prioritise_search and self.n > 1) else None
                                for _ in range(self.n)]
            from speechbrain_amd import native

            for st in list(self.enc_streams) + [d for d in self.dec_streams if d is not None]:
                native.retain_stream_workspace(st)  # (released in close(); pooled handles may be shared with another owner)
        self._closed = False
        self.pool = ThreadPoolExecutor(self.n)
        self._take_lock = threading.Lock()
        self.balance_tail = True
        self.plan_workers = self.n  # workers the group sizes are planned for (a one-worker replay of an eight-worker job sets 8)

    def close(self):
        """Stop the worker threads and return what the streams of this transcriber pinned in the library (one registered
        workspace per stream that issued an op: native.release_stream_workspace).  The object must not be used afterwards."""
        self.pool.shutdown(wait=True)
        if self.device.type == "cuda" and not self._closed:
            self._closed = True
            from speechbrain_amd import native

            streams = list(self.enc_streams) + [d for d in self.dec_streams if d is not None]
            native.release_search_workspaces(streams)  # the searches' grow-only buffers (several GB each at 128-utterance groups)
            for st in streams:
                native.release_stream_workspace(st)

    def __del__(self):  # (a transcriber dropped without close(): its streams' search buffers must not outlive it)
        try:
            if self.device.type == "cuda" and not self._closed:
                from speechbrain_amd import native

                streams = list(self.enc_streams) + [d for d in self.dec_streams if d is not None]
                native.release_search_workspaces(streams)
                for st in streams:
                    native.forget_stream_owner(st)  # (no synchronisation in a finaliser: the claim goes, the workspace stays)
        except Exception:
            pass

    def _one(self, slot: int, wavs, wav_lens, prepare: Optional[Callable], ready: Optional[Callable] = None):
        searcher = self.searchers[slot]
        if prepare is not None:
            prepare(searcher, wavs)
        if ready is not None:
            ready()  # the batch may still be on its way (ShardedTranscriber.distribute): order this stream after it
        with torch.no_grad():
            # host batches (pinned memory: the copy is asynchronous) go to the device on THIS worker's stream, so
            # the transfer of one batch overlaps the kernels of the batches in flight on the other streams
            wavs = wavs.to(self.device, non_blocking=True)
            wav_lens = wav_lens.to(self.device, non_blocking=True)
            if wavs.dtype == torch.int16:  # PCM as shipped by ShardedTranscriber.scatter / read from a wav file
                from speechbrain_amd import native

                wavs = native.pcm16_to_f32(wavs)
            enc = self.asr.encode_batch(wavs, wav_lens)
            dec_stream = self.dec_streams[slot] if self.device.type == "cuda" else None
            if dec_stream is None:
                toks, _, _, _ = searcher(enc, wav_lens)
            else:
                cur = torch.cuda.current_stream()
                dec_stream.wait_stream(cur)
                with torch.cuda.stream(dec_stream):
                    toks, _, _, _ = searcher(enc, wav_lens)  # returns host token lists: dec_stream is drained
                enc.record_stream(dec_stream)
                cur.wait_stream(dec_stream)
        return toks

    def _many(self, slot: int, ks, batches, prepare: Optional[Callable], ready=None):
        """Encode the batches `ks` together (or one after the other), decode them in one grouped search."""
        searcher = self.searchers[slot]
        items, ratios, dev_batches = [], [], []
        with torch.no_grad():
            for k in ks:
                wavs, wav_lens = batches[k]
                if prepare is not None:
                    prepare(searcher, wavs)
                ratios.append((searcher.min_decode_ratio, searcher.max_decode_ratio))
                if ready is not None and ready[k] is not None:
                    ready[k]()
                wavs = wavs.to(self.device, non_blocking=True)
                wav_lens = wav_lens.to(self.device, non_blocking=True)
                if wavs.dtype == torch.int16:
                    from speechbrain_amd import native

                    wavs = native.pcm16_to_f32(wavs)
                dev_batches.append((wavs, wav_lens))
            if self.group_encoder and hasattr(self.asr, "encode_group"):
                # one encoder pass over the rows of several batches (True: all of the group's; an int: that many at a time)
                per = len(dev_batches) if self.group_encoder is True else max(1, int(self.group_encoder))
                encs = []
                for i in range(0, len(dev_batches), per):
                    encs += self.asr.encode_group(dev_batches[i: i + per])
            else:
                encs = [self.asr.encode_batch(w, l) for w, l in dev_batches]
            items = [(e, l) for e, (_, l) in zip(encs, dev_batches)]
            dec_stream = self.dec_streams[slot] if self.device.type == "cuda" else None
            if dec_stream is None:
                res = searcher.forward_group(items, ratios)
            else:
                cur = torch.cuda.current_stream()
                dec_stream.wait_stream(cur)
                with torch.cuda.stream(dec_stream):
                    res = searcher.forward_group(items, ratios)  # returns host token lists: dec_stream is drained
                for enc, _ in items:
                    enc.record_stream(dec_stream)
                cur.wait_stream(dec_stream)
        return [(k, r[0]) for k, r in zip(ks, res)]

    def _work(self, slot: int, todo: "queue.Queue", batches, prepare, ready=None):
        out = []

        def take():
            # the whole group under one lock: groups are the same runs of the queue order however the workers
            # interleave, so a grouped search sees the same rows (-> the same kernels, bit-identical results) in
            # every run of the same job, with one worker or with eight
            ks = []
            with self._take_lock:
                # towards the end of the job the groups shrink so that every worker still gets one (the last round of
                # an 80-batch job would otherwise keep half of the workers idle); the sizes depend only on the number of
                # batches left, not on who asks
                want = self.group
                if self.balance_tail:
                    want = max(1, min(self.group, -(-todo.qsize() // self.plan_workers)))
                while len(ks) < want:
                    try:
                        ks.append(todo.get_nowait())
                    except queue.Empty:
                        break
            return ks

        def run(ks):
            if len(ks) > 1 and hasattr(self.searchers[slot], "forward_group"):
                out.extend(self._many(slot, ks, batches, prepare, ready))
            else:
                out.extend((k, self._one(slot, *batches[k], prepare, ready[k] if ready is not None else None)) for k in ks)

        if self.device.type != "cuda":
            while True:
                ks = take()
                if not ks:
                    return out
                run(ks)
        torch.cuda.set_device(self.device)  # the current device is per host thread
        with torch.cuda.stream(self.enc_streams[slot]):
            while True:
                ks = take()
                if not ks:
                    break
                run(ks)
            self.enc_streams[slot].synchronize()
        # (the grow-only search buffers belong to the worker STREAMS and stay for the transcriber's next job: close() returns them)
        return out

    def transcribe_batches(self, batches: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                           prepare: Optional[Callable] = None, ready: Optional[Sequence] = None) -> List[list]:
        """Workers pull from one queue, largest batch first, so that they finish together.  ``ready[k]`` (optional):
        called by the worker, on its stream, before it touches batch k (a batch that is still being received)."""
        return self.finish(self.start(batches, prepare, ready))

    def finish(self, running) -> List[list]:
        futs, n = running
        res = {}
        for f in futs:
            for k, toks in f.result():
                res[k] = toks
        return [res[k] for k in range(n)]

    def start(self, batches: Sequence[Tuple[torch.Tensor, torch.Tensor]], prepare: Optional[Callable] = None,
              ready: Optional[Sequence] = None):
        """``transcribe_batches`` without the wait: the workers are running when this returns (``finish`` collects).
        The caller may go on producing the batches meanwhile -- a worker blocks in ``ready[k]()`` until batch k is there."""
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            for s in self.enc_streams:
                s.wait_stream(cur)
        todo: "queue.Queue" = queue.Queue()
        for k in sorted(range(len(batches)), key=lambda i: -batches[i][0].numel()):
            todo.put(k)
        n_workers = min(self.n, (len(batches) + self.group - 1) // self.group)
        futs = [self.pool.submit(self._work, slot, todo, batches, prepare, ready) for slot in range(max(1, n_workers))]
        return futs, len(batches)
