"""Single-utterance latency (B = 1, 10 s, 40 decoding steps, beam 10 + CTC: the bench's p50 case) with the decode step as the
persistent few-row launch (csrc/decoder_persist.hip, knob 47) or as launches per operation -- wall clock per call, the search
alone from a fixed encoder output, and (--report) HIP-event time per kernel class.

    python tools/latency_probe.py [--runs 7] [--knob 47=0] [--knob 48=64] [--overlap 0|3] [--report]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speechbrain_amd import native
from speechbrain_amd.inference.builders import build_asr

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=7)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--overlap", type=int, default=0)
ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--report", action="store_true")
ap.add_argument("--stamps", action="store_true", help="phase times of the last persistent step of a search (knob 49)")
ap.add_argument("--knob", action="append", default=[])
args = ap.parse_args()

dev = torch.device("cuda:0")
lib = native.load()
for kv in args.knob:
    lib.sbk_prof_set_knob(*[int(v) for v in kv.split("=")])
asr = build_asr("L", vocab=5000, seed=0, beam_size=10, ctc_weight=0.4, device="cuda:0")
dec = asr.mods.decoder
dec.overlap_ctc, dec.graph_mode = args.overlap, args.graph
n = int(args.seconds * 16000)
wav = (0.1 * torch.randn(1, n, generator=torch.Generator().manual_seed(5))).pin_memory()
lens = torch.ones(1)
frames = ((1 + n // 160 - 1) // 2 + 1 - 1) // 2 + 1
steps = int(round(4 * args.seconds))
dec.max_decode_ratio = (steps + 0.5) / frames
st = torch.cuda.Stream(dev)
with torch.no_grad(), torch.cuda.stream(st):
    def med(fn):
        fn()
        ts = []
        for _ in range(args.runs):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        ts.sort()
        return 1e3 * ts[len(ts) // 2]

    whole = med(lambda: asr.transcribe_batch(wav.to(dev, non_blocking=True), lens))
    enc = asr.encode_batch(wav.to(dev), lens)
    ldev = lens.to(dev)
    enc_ms = med(lambda: asr.encode_batch(wav.to(dev, non_blocking=True), lens))
    search = med(lambda: dec(enc, ldev))
    print(f"latency probe: {args.seconds:g} s utterance, {steps} steps: transcribe_batch {whole:.2f} ms = encoder {enc_ms:.2f} + search {search:.2f} "
          f"({1e3 * search / steps:.1f} us per step); knobs {args.knob}, overlap_ctc {args.overlap}, graph {args.graph}", flush=True)
    if args.report:
        native.prof_reset()
        native.prof_enable(True)
        dec(enc, ldev)
        torch.cuda.synchronize()
        native.prof_enable(False)
        rep = native.prof_report()
        tot = sum(v["ms"] for v in rep.values())
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
            print(f"  {k:24s} {v['count']:6d} launches {1e3 * v['ms'] / v['count']:8.1f} us each {100 * v['ms'] / tot:5.1f} %")
    if args.stamps:
        import ctypes

        lib.sbk_prof_set_knob(49, 1)
        dec(enc, ldev)
        torch.cuda.synchronize()
        lib.sbk_prof_set_knob(49, 0)
        buf = (ctypes.c_longlong * 256)()
        n = lib.sbk_prof_persist_stamps(buf, 256)
        t = [buf[i] * 0.01 for i in range(n)]  # 100 MHz ticks -> us
        names = ["embed"]
        for l in range(6):
            names += [f"L{l} norm1+in_proj", f"L{l} self-attn", f"L{l} out_proj", f"L{l} norm2+q", f"L{l} cross-attn", f"L{l} out_proj2", f"L{l} norm3+ffn0", f"L{l} ffn3"]
        print(f"  persistent step (last of the search): {t[-1] - t[0]:.1f} us, {n} stamps")
        work = {}
        bars = 0.0
        for k, name in enumerate(names):
            w, b = t[1 + 2 * k] - t[2 * k], t[2 + 2 * k] - t[1 + 2 * k]
            key = name.split(" ", 1)[-1] if name != "embed" else name
            work.setdefault(key, []).append(w)
            bars += b
        for key, v in work.items():
            print(f"    {key:16s} {sum(v) / len(v):7.2f} us each x {len(v)}")
        print(f"    barriers        {bars / len(names):7.2f} us each x {len(names)} (arrival .. release, workgroup 0)")
        print(f"    norm+seq_lin    {t[-1] - t[-2]:7.2f} us")
