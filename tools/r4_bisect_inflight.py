"""Round 4, first GPU visit: why do batches in flight differ from sequential transcribe_batch calls
(tests/test_full_size_gpu.py::test_batches_in_flight_match_sequential_conformer_l, red in GPUTEST_r03)?

One process, one model; every leg prints which batches differ from the reference leg and by how many tokens.
Discriminators: sequential determinism, overlap_ctc on the sequential side, the split-operand
route off, one worker, workers without the high-priority search stream, worker streams warmed first."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from speechbrain_amd import native  # noqa: E402
from speechbrain_amd.inference.builders import build_asr  # noqa: E402
from speechbrain_amd.inference.streams import ConcurrentTranscriber  # noqa: E402


def diff(a, b):
    out = []
    for k, (x, y) in enumerate(zip(a, b)):
        bad = sum(1 for u, v in zip(x, y) if u != v)
        ntok = sum(sum(1 for p, q in zip(u, v) if p != q) + abs(len(u) - len(v)) for u, v in zip(x, y))
        if bad:
            out.append((k, bad, ntok))
    return out


def main():
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

    def conc(streams=6, prio=True, ov=None):
        ct = ConcurrentTranscriber(asr, streams=streams, prioritise_search=prio)
        if ov is not None:
            for s in ct.searchers:
                s.overlap_ctc = ov
        return ct.transcribe_batches(batches, prepare=fix_len)

    def leg(name, fn, ref):
        t = time.time()
        try:
            r = fn()
            print(f"{name:58s} {time.time() - t:5.1f}s  differs: {diff(ref, r) if ref is not None else '-'}", flush=True)
            return r
        except Exception as e:  # noqa: BLE001
            print(f"{name:58s} FAILED {type(e).__name__}: {e}", flush=True)
            return None

    ref = leg("A  sequential, overlap_ctc 3 (the test's reference)", seq, None)
    leg("A2 sequential again", seq, ref)
    leg("B  sequential, overlap_ctc 0", lambda: seq(0), ref)
    leg("C  6 workers (the test)", conc, ref)
    leg("C2 6 workers again", conc, ref)
    leg("D  1 worker (worker machinery, overlap 3)", lambda: conc(1), ref)
    leg("D2 1 worker, overlap 0", lambda: conc(1, ov=0), ref)
    leg("E  6 workers, search on the encoder stream", lambda: conc(6, prio=False), ref)
    leg("F  2 workers", lambda: conc(2), ref)
    native.F32X3 = False
    ref0 = leg("G  sequential, F32X3 off (python route only)", seq, ref)
    leg("H  6 workers, F32X3 off (vs G)", conc, ref0)
    native.F32X3 = True
    lib = native.load()
    # the library's own split-operand routes (memory / CTC / vocabulary projections of the search): off as well
    native.F32X3 = False
    lib.sbk_prof_set_knob(34, 1 << 30)
    r0 = leg("I  sequential, no split-operand kernel anywhere", seq, ref)
    leg("J  6 workers, no split-operand kernel anywhere (vs I)", conc, r0)
    lib.sbk_prof_set_knob(18, 0)  # and no persistent / stream-K kernel either (tile-grid kernels only)
    r1 = leg("L  sequential, tile-grid GEMMs only", seq, ref)
    leg("M  6 workers, tile-grid GEMMs only (vs L)", conc, r1)
    lib.sbk_prof_set_knob(18, 1)
    lib.sbk_prof_set_knob(34, 1024)
    native.F32X3 = True
    leg("K  6 workers, last", conc, ref)


if __name__ == "__main__":
    main()
