"""Secondary measurements (not the bench.py contract): BASELINE.json configs[1] (Conformer-S encoder, 32 x 10 s)
and the worst case of configs[2] (max_decode_ratio = 1.0: one decoding step per encoder frame), SURVEY 8(d)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from speechbrain_amd import native
from speechbrain_amd.inference.builders import build_asr

dev = torch.device("cuda:0")
native.load()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def encoder_s():
    asr = build_asr("S", vocab=5000, seed=0, device=str(dev))
    wav = (0.1 * torch.randn(32, 160000, generator=torch.Generator().manual_seed(1234))).to(dev)
    out = {}
    for tag, lens in (("full", torch.ones(32)), ("ragged", torch.linspace(0.5, 1.0, 32))):
        lens = lens.to(dev)
        dt = timed(lambda: asr.encode_batch(wav, lens), 10)
        out[tag] = {"ms": round(dt * 1e3, 2), "audio_sec_per_s": round(float((lens * 10.0).sum()) / dt, 1)}
    print(json.dumps({"workload": "configs[1]: STFT+Fbank+CNN+Conformer-S encoder, 32 x 10 s, fp32", **out}), flush=True)


def worst_case():
    asr = build_asr("L", vocab=5000, seed=0, beam_size=10, ctc_weight=0.4, device=str(dev))
    asr.mods.decoder.check_every = 0
    (wav, lens, secs), = bench.make_batches(1, 32, seed=1234)
    wav, lens = wav.to(dev), lens.to(dev)
    asr.mods.decoder.max_decode_ratio = 1.0

    def go():
        asr.transcribe_batch(wav, lens)
    dt = timed(go, 2)
    T = bench.frames_after_frontend(wav.shape[1])
    print(json.dumps({"workload": "configs[2] worst case: Conformer-L beam 10 + CTC, 32 utterances U(5,30) s, "
                                  "max_decode_ratio 1.0", "decode_steps": T, "seconds_per_batch": round(dt, 3),
                      "audio_sec_per_s": round(sum(secs) / dt, 1)}), flush=True)


if __name__ == "__main__":
    encoder_s()
    worst_case()
