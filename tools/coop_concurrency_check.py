"""Several host threads, each on its own stream, run single-utterance searches (the persistent cooperative step of
csrc/decoder_persist.hip) at the same time, next to a thread that keeps the chip busy with large contractions: results must
equal the sequential ones and nothing may hang.  Run under `timeout`: a hang is the finding.

    timeout 120 python tools/coop_concurrency_check.py [--threads 4] [--rounds 12]
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speechbrain_amd import native
from speechbrain_amd.inference.builders import build_asr

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=4)
ap.add_argument("--rounds", type=int, default=12)
args = ap.parse_args()
dev = torch.device("cuda:0")
native.load()
asr = build_asr("L", vocab=5000, seed=0, beam_size=10, ctc_weight=0.4, device="cuda:0")
with torch.no_grad():
    asr.mods.seq_lin.w.weight.mul_(8.0)
    asr.mods.ctc_lin.w.weight.mul_(8.0)
dec = asr.mods.decoder
g = torch.Generator().manual_seed(1)
encs = [(torch.randn(1, 120 + 17 * k, 512, generator=g).to(dev), torch.ones(1, device=dev)) for k in range(args.threads)]
dec.max_decode_ratio = 0.1
with torch.no_grad():
    ref = [dec(e, l)[0] for e, l in encs]
torch.cuda.synchronize()
rep_ok = True
native.prof_reset(); native.prof_enable(True)
with torch.no_grad():
    dec(*encs[0])
native.prof_enable(False)
assert "decoder_step_persist" in native.prof_report(), "the searches of this check must take the persistent step"
stop = threading.Event()
bad = []


def load():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    a = torch.randn(8192, 512, device=dev)
    w = torch.randn(2048, 512, device=dev)
    with torch.cuda.stream(s), torch.no_grad():
        while not stop.is_set():
            for _ in range(4):
                native.gemm_nt(a, w)
            s.synchronize()


def worker(k):
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        for r in range(args.rounds):
            got = dec(*encs[k])[0]
            if got != ref[k]:
                bad.append((k, r))
        s.synchronize()


t0 = time.perf_counter()
lt = threading.Thread(target=load)
lt.start()
ths = [threading.Thread(target=worker, args=(k,)) for k in range(args.threads)]
for t in ths:
    t.start()
for t in ths:
    t.join()
stop.set()
lt.join()
torch.cuda.synchronize()
print(f"coop concurrency check: {args.threads} threads x {args.rounds} single-utterance searches beside a contraction load: "
      f"{len(bad)} differing results, {time.perf_counter() - t0:.2f} s", flush=True)
sys.exit(1 if bad else 0)
