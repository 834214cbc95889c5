"""Per-kernel time of a Whisper large-v3-shape encoder forward (8 x 30 s, `--layers` of the 32 layers) per precision.

    python tools/whisper_probe.py [--layers 4] [--prec fp32,bf16]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speechbrain_amd import native
from speechbrain_amd.integrations.huggingface.whisper import Whisper

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--prec", default="fp32,bf16")
ap.add_argument("--knob", action="append", default=[], metavar="KEY=VALUE")
args = ap.parse_args()
dev = torch.device("cuda:0")
for kv in args.knob:
    native.load().sbk_prof_set_knob(*[int(v) for v in kv.split("=")])
cfg = dict(num_mel_bins=128, d_model=1280, encoder_layers=args.layers, encoder_attention_heads=20, encoder_ffn_dim=5120,
           max_source_positions=1500, decoder_layers=0, decoder_attention_heads=20, decoder_ffn_dim=5120,
           vocab_size=51866, max_target_positions=448)
w = Whisper.from_config(cfg, encoder_only=True).to(dev).eval()
mel = w._get_mel((0.1 * torch.randn(args.batch, 480000, generator=torch.Generator().manual_seed(3))).to(dev))
ref = None
for prec in args.prec.split(","):
    with native.precision_scope(prec), torch.no_grad():
        out = w.forward_encoder(mel)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            w.forward_encoder(mel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        native.prof_reset()
        native.prof_enable(True)
        w.forward_encoder(mel)
        torch.cuda.synchronize()
        native.prof_enable(False)
    rep = native.prof_report()
    tot = sum(v["ms"] for v in rep.values())
    if ref is None:
        ref = out.float()
    err = float((out.float() - ref).abs().max()), float((out.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"{prec}: {1e3 * dt:.2f} ms per forward ({args.layers} layers, {args.batch} x 30 s); kernel events {tot:.2f} ms; "
          f"max |d| vs first precision {err[0]:.3e}, relative rms {err[1]:.3e}", flush=True)
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:10]:
        print(f"  {k:22s} {v['count']:5d} launches {1e3 * v['ms'] / v['count']:9.1f} us each {100 * v['ms'] / tot:5.1f} %  "
              f"{v['bytes'] / v['ms'] / 1e6:8.1f} GB/s {v['flops'] / v['ms'] / 1e9:8.1f} TF/s")
