"""A short grouped beam search at the bench's decode shape (4 batches x 32 utterances, T' ~ 430, beam 10 + CTC), fed
with a random encoder output -- the decode-step kernels alone, for rocprofv3 --pmc / --kernel-trace passes.

    python tools/decode_probe.py [--steps 12] [--reps 1] [--report]
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
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--frames", type=int, default=430)
ap.add_argument("--batches", type=int, default=4)
ap.add_argument("--utts", type=int, default=32, help="utterances per batch")
ap.add_argument("--report", action="store_true", help="HIP-event time per kernel class (sbk_prof_*)")
ap.add_argument("--knob", action="append", default=[])
args = ap.parse_args()

dev = torch.device("cuda:0")
native.load()
for kv in args.knob:
    native.load().sbk_prof_set_knob(*[int(v) for v in kv.split("=")])
asr = build_asr("L", vocab=5000, seed=0, beam_size=10, ctc_weight=0.4, device="cuda:0")
dec = asr.mods.decoder
g = torch.Generator().manual_seed(3)
items, ratios = [], []
for k in range(args.batches):
    T = args.frames - 10 * k
    enc = torch.randn(args.utts, T, 512, generator=g).to(dev)
    lens = torch.linspace(0.85, 1.0, args.utts).to(dev)
    items.append((enc, lens))
    ratios.append((0.0, (args.steps + 0.5) / T))
with torch.no_grad():
    dec.forward_group(items, ratios)  # warm-up (workspaces, handles)
    torch.cuda.synchronize()
    if args.report:
        native.prof_reset()
        native.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        dec.forward_group(items, ratios)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"decode probe: {args.batches} x {args.utts} utterances, T' {args.frames}, {args.steps} steps: {1e3 * dt / args.reps / args.steps:.3f} ms per step", flush=True)
if args.report:
    native.prof_enable(False)
    rep = native.prof_report()
    tot = sum(v["ms"] for v in rep.values())
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
        per = 1e3 * v["ms"] / v["count"]
        print(f"  {k:20s} {v['count']:6d} launches {per:8.1f} us each {100 * v['ms'] / tot:5.1f} %  {v['bytes'] / v['ms'] / 1e6:8.1f} GB/s {v['flops'] / v['ms'] / 1e9:7.1f} TF/s")
