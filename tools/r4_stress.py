"""Round 4, visit B: the in-flight test was red ONCE (GPUTEST_r03) and green in every other run -- a rare race or a read
of uninitialised memory.  Two instruments:

  --poison   every torch.empty / empty_like on the GPU is filled with a byte pattern first (0xFF = NaN / -1, 0x00); the
             token ids of sequential and in-flight runs must not depend on the pattern (a read of memory nobody wrote
             shows up as a difference -- deterministically, no luck needed)
  --stress N the in-flight leg N times against ONE sequential reference; then the sequential leg N times
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from speechbrain_amd import native  # noqa: E402
from speechbrain_amd.inference.builders import build_asr  # noqa: E402
from speechbrain_amd.inference.streams import ConcurrentTranscriber  # noqa: E402

_real_empty, _real_empty_like = torch.empty, torch.empty_like
_POISON = [None]


def _fill(t):
    v = _POISON[0]
    if v is not None and t.is_cuda and t.numel() and t.is_contiguous():
        t.reshape(-1).view(torch.uint8).fill_(v)
    return t


def _empty(*a, **k):
    return _fill(_real_empty(*a, **k))


def _empty_like(*a, **k):
    return _fill(_real_empty_like(*a, **k))


torch.empty, torch.empty_like = _empty, _empty_like


def diff(a, b):
    out = []
    for k, (x, y) in enumerate(zip(a, b)):
        bad = sum(1 for u, v in zip(x, y) if u != v)
        if bad:
            out.append((k, bad))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poison", action="store_true")
    ap.add_argument("--stress", type=int, default=0)
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--group", type=int, default=1)
    args = ap.parse_args()
    native.load()
    asr = build_asr("L", device="cuda:0", beam_size=10, ctc_weight=0.4)
    asr.mods.decoder.check_every = 0
    g = torch.Generator().manual_seed(33)
    batches = []
    for k, (B, sec) in enumerate([(24, 9.0), (16, 14.0), (32, 6.0), (8, 20.0), (16, 11.0), (24, 7.5), (12, 16.0)]):
        n = int(sec * 16000)
        wav = 0.1 * torch.randn(B, n, generator=g)
        lens = torch.linspace(0.6, 1.0, B)
        for i in range(B):
            wav[i, int(lens[i] * n):] = 0
        batches.append((wav.cuda(), lens.cuda()))

    def fix_len(searcher, wavs):
        T = ((1 + wavs.shape[1] // 160 - 1) // 2 + 1 - 1) // 2 + 1
        searcher.max_decode_ratio = 20.5 / T

    def seq(overlap=3):
        asr.mods.decoder.overlap_ctc = overlap
        out = []
        for w, l in batches:
            fix_len(asr.mods.decoder, w)
            out.append(asr.transcribe_batch(w, l)[1])
        asr.mods.decoder.overlap_ctc = 3
        return out

    def conc(streams=None, group=None):
        ct = ConcurrentTranscriber(asr, streams=streams or args.workers, group=group or args.group)
        return ct.transcribe_batches(batches, prepare=fix_len)

    ref = seq()
    print("reference: sequential, natural memory", flush=True)
    if args.poison:
        for pat in (0xFF, 0x00, 0xFF):
            _POISON[0] = pat
            torch.cuda.synchronize()
            for name, fn in (("sequential overlap 3", seq), ("sequential overlap 0", lambda: seq(0)), ("6 in flight", conc),
                             ("3 in flight, groups of 2", lambda: conc(3, 2))):
                t = time.time()
                r = fn()
                print(f"poison {pat:#04x}  {name:28s} {time.time() - t:5.1f}s  differs: {diff(ref, r)}", flush=True)
        _POISON[0] = None
        # the encoder alone (its output tensor compared bit for bit)
        w, l = batches[5]
        e0 = asr.encode_batch(w, l).clone()
        for pat in (0xFF, 0x00):
            _POISON[0] = pat
            e1 = asr.encode_batch(w, l)
            d = (e1 != e0) & ~(torch.isnan(e1) & torch.isnan(e0))
            print(f"poison {pat:#04x}  encoder batch 5: {int(d.sum())} elements differ, nan {int(torch.isnan(e1).sum())}", flush=True)
        _POISON[0] = None
    if args.stress:
        bad = []
        t = time.time()
        for i in range(args.stress):
            r = conc()
            d = diff(ref, r)
            if d:
                bad.append((i, d))
                print(f"  in-flight rep {i}: differs {d}", flush=True)
        print(f"stress: {args.workers} in flight x {args.stress}: {len(bad)} differing reps, {time.time() - t:.1f}s", flush=True)
        bad = []
        t = time.time()
        for i in range(args.stress):
            r = seq()
            d = diff(ref, r)
            if d:
                bad.append((i, d))
                print(f"  sequential rep {i}: differs {d}", flush=True)
        print(f"stress: sequential x {args.stress}: {len(bad)} differing reps, {time.time() - t:.1f}s", flush=True)
        # one long-lived transcriber (the bench's mode): the same worker streams again and again
        ct = ConcurrentTranscriber(asr, streams=args.workers)
        bad = 0
        for i in range(args.stress):
            d = diff(ref, ct.transcribe_batches(batches, prepare=fix_len))
            if d:
                bad += 1
                print(f"  long-lived rep {i}: differs {d}", flush=True)
        print(f"stress: one long-lived transcriber x {args.stress}: {bad} differing reps", flush=True)


if __name__ == "__main__":
    main()
