#!/usr/bin/env python
"""bench.py -- audio-sec/s of the EncoderDecoderASR hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 16 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], "Full Conformer-L enc-dec + S2STransformerBeamSearcher
beam=10, LibriSpeech-shape synthetic"): Conformer-L (d 512, 12 enc / 6 dec layers, 5000 tokens,
RelPosMHAXL), beam 10 + CTC weight 0.4 (the recipe's valid_search), seeded random weights,
synthetic 16 kHz audio 0.1*randn, utterance durations U(5,30) s (seed 1234), duration-sorted
batches of 128 utterances (sized for 288 GB of HBM; --batch 32 gives the recipe-sized batches),
--streams batches in flight through speechbrain_amd.inference.streams.ConcurrentTranscriber (one
host thread per batch in flight, the encoder on a normal- and the search on a high-priority HIP
stream).  One "step" = one batch through Fbank -> norm -> CNN -> Conformer encoder -> beam search ->
token ids on the host.  Random weights never emit EOS, so the number of decoding steps is fixed
through max_decode_ratio to round(4 tokens/s * seconds).
Waveforms are resident in HBM when the timed region starts.  fp32 arithmetic throughout.

One JSON line on rank 0: metric / value (total unpadded audio seconds of all ranks / max-over-ranks
wall time) plus "roofline" (dominant kernel, from HIP-event timing of every launch during an
instrumented repetition of the same steps) and "cpu_baseline" (the oracle port of the reference's
PyTorch-CPU path, timed here on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
MFMA_KERNELS = ("gemm", "relpos_attention")
TOKENS_PER_SECOND = 4.0


def make_batches(n_batches, batch, seed, sr=16000, lo=5.0, hi=30.0):
    """Duration-sorted batches of synthetic audio: list of (wavs [B,N] cpu, rel_lens [B], seconds list)."""
    g = torch.Generator().manual_seed(seed)
    dur = (lo + (hi - lo) * torch.rand(n_batches * batch, generator=g)).sort().values
    out = []
    for i in range(n_batches):
        d = dur[i * batch:(i + 1) * batch]
        n = (d * sr).round().long()
        N = int(n.max())
        wav = 0.1 * torch.randn(batch, N, generator=g)
        for r in range(batch):
            wav[r, int(n[r]):] = 0.0
        out.append((wav, n.float() / N, (n.float() / sr).tolist()))
    return out


def frames_after_frontend(n_samples):
    t = 1 + n_samples // 160
    t = (t - 1) // 2 + 1
    return (t - 1) // 2 + 1


def set_decode_steps(asr, n_samples):
    """Fix the decode length to round(4 tok/s * padded seconds) through max_decode_ratio."""
    T = frames_after_frontend(n_samples)
    steps = max(1, int(round(TOKENS_PER_SECOND * n_samples / 16000.0)))
    asr.mods.decoder.max_decode_ratio = (steps + 0.5) / T
    return steps


def run_step(asr, wav, lens):
    """One batch through the whole path on the current stream (latency case, instrumented pass)."""
    set_decode_steps(asr, wav.shape[1])
    words, toks = asr.transcribe_batch(wav, lens)
    return toks


def fixed_decode_length(searcher, wavs):
    """`prepare` hook of ConcurrentTranscriber: decode steps = round(4 tok/s * padded seconds)."""
    T = frames_after_frontend(wavs.shape[1])
    steps = max(1, int(round(TOKENS_PER_SECOND * wavs.shape[1] / 16000.0)))
    searcher.max_decode_ratio = (steps + 0.5) / T


def cpu_threads():
    """Host threads for the CPU leg: the cores this process may run on, capped at 32 (the port's
    small per-frame ops stop scaling long before that and oversubscription only hurts)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def cpu_baseline_subprocess(timeout_s=240):
    """Run the CPU leg in a child process so that a slow host can never stall the GPU result."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=timeout_s)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "audio-sec/s", "cores": cpu_threads(), "kind": "port",
                "sample": "CPU leg failed: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "audio-sec/s", "cores": cpu_threads(), "kind": "port",
                "sample": f"CPU leg did not finish within {timeout_s} s"}


def cpu_baseline(asr, seconds=6.0, batch=2):
    """Oracle port of the reference's CPU path (no KV cache, Python-loop CTC scorer) on a bounded sample."""
    from oracle import sb_oracle as O
    from speechbrain_amd.inference.builders import flat_state_dict

    torch.set_num_threads(cpu_threads())
    sd = flat_state_dict(asr)
    fc = O.FbankCfg(n_fft=512, n_mels=80, win_length_ms=32)
    mc = O.ModelCfg()
    n = int(seconds * 16000)
    wav = 0.1 * torch.randn(batch, n, generator=torch.Generator().manual_seed(99))
    lens = torch.ones(batch)
    T = frames_after_frontend(n)
    steps = max(1, int(round(TOKENS_PER_SECOND * seconds)))
    sc = O.SearchCfg(beam=10, ctc_weight=0.4, max_decode_ratio=(steps + 0.5) / T)
    t0 = time.time()
    with torch.no_grad():
        enc = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
        O.beam_search(enc, lens, sd, mc, sc)
    dt = time.time() - t0
    return {"value": round(batch * seconds / dt, 3), "unit": "audio-sec/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{batch} x {seconds:g} s utterance, Conformer-L beam 10 + CTC 0.4, {steps} decode steps, "
                                      f"oracle/sb_oracle.py (torch-CPU fp32) in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="utterances per duration-sorted batch (32 = the recipe-sized batches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--latency-runs", type=int, default=5)
    ap.add_argument("--streams", type=int, default=8, help="independent batches in flight per GPU")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--attention", default="RelPosMHAXL", choices=["RelPosMHAXL", "RoPEMHA"],
                    help="encoder attention (RelPosMHAXL = BASELINE.json's config; RoPEMHA = the in-tree recipe)")
    ap.add_argument("--lm", action="store_true",
                    help="add the recipe's TransformerLM scorer (12 x 768, weight 0.6, T=1.15): test_search at beam 10")
    ap.add_argument("--no-search-priority", action="store_true",
                    help="run each worker's search on its normal-priority stream (A/B of the stream priorities)")
    ap.add_argument("--knob", action="append", default=[], metavar="KEY=VALUE",
                    help="tuning switch passed to sbk_prof_set_knob (A/B measurements only)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    def note(msg):
        if args.verbose:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    if args.cpu_baseline_only:
        from speechbrain_amd.inference.builders import build_asr

        print(json.dumps(cpu_baseline(build_asr("L", vocab=5000, seed=0, device="cpu"))), flush=True)
        return

    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SBK_BENCH_FORCE_DIST=1 (with torchrun --nproc-per-node 1): run the RCCL leg -- process group, gather,
    # all-reduce, barrier -- on a single GPU, to check the N > 1 code path where only one GPU is available
    dist_on = world > 1 or (os.environ.get("SBK_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if dist_on:
        dist.init_process_group("nccl", device_id=dev)

    from speechbrain_amd import native
    from speechbrain_amd.inference.builders import build_asr

    native.load()
    for kv in args.knob:
        native.load().sbk_prof_set_knob(*[int(v) for v in kv.split("=")])
    asr = build_asr("L", vocab=5000, seed=0, beam_size=10, ctc_weight=0.4, device=str(dev),
                    attention_type=args.attention)
    if args.lm:  # conformer_large.yaml:166-223: full_scorers=[transformerlm, ctc], lm_weight 0.6, temperature 1.15
        from speechbrain_amd.decoders import (CTCScorer, S2STransformerBeamSearcher, ScorerBuilder,
                                              TransformerLMScorer)
        from speechbrain_amd.lobes.models.transformer.TransformerLM import TransformerLM

        torch.manual_seed(1)
        lm = TransformerLM(vocab=5000, d_model=768, nhead=12, num_encoder_layers=12, num_decoder_layers=0, d_ffn=3072,
                           dropout=0.0, activation=torch.nn.GELU, normalize_before=False).to(dev).eval()
        scorer = ScorerBuilder(full_scorers=[TransformerLMScorer(language_model=lm, temperature=1.15),
                                             CTCScorer(ctc_fc=asr.mods.ctc_lin, blank_index=0, eos_index=2)],
                               weights={"transformerlm": 0.6, "ctc": 0.4})
        asr.mods.decoder = S2STransformerBeamSearcher(
            modules=[asr.mods.transformer, asr.mods.seq_lin], bos_index=1, eos_index=2, min_decode_ratio=0.0,
            max_decode_ratio=1.0, beam_size=10, using_eos_threshold=False, length_normalization=True,
            temperature=1.15, scorer=scorer)
    asr.mods.decoder.check_every = 0  # fixed-length decoding: no stop-rule polling, fully asynchronous

    # every rank owns K (+W) batches: weak scaling, per-GPU work fixed as N grows
    pool = make_batches(args.steps, args.batch, seed=1234 + rank)
    pool_dev = [(w.to(dev), l.to(dev), s) for w, l, s in pool]
    warm_dev = [pool_dev[-1]] * args.warmup  # the longest batch: sizes every allocation before the timed region
    audio_sec = sum(sum(s) for _, _, s in pool)

    note(f"model built; {args.steps} batches resident; warm-up")
    from speechbrain_amd.inference.streams import ConcurrentTranscriber

    workers = ConcurrentTranscriber(asr, streams=max(1, args.streams), prioritise_search=not args.no_search_priority)
    for w, l, _ in warm_dev:  # every worker stream sizes its allocations on the longest batch
        workers.transcribe_batches([(w, l)] * workers.n, prepare=fixed_decode_length)
    note("timed region")

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    results = []
    for k, hyps in enumerate(workers.transcribe_batches([(w, l) for w, l, _ in pool_dev], prepare=fixed_decode_length)):
        results.extend((k * args.batch + i, h) for i, h in enumerate(hyps))
    if dist_on:  # token ids to rank 0: the path's only collective
        width = int(round(TOKENS_PER_SECOND * 30.0)) + 8  # same shape on every rank (durations <= 30 s)
        host = torch.tensor([[len(h)] + list(h) + [0] * (width - len(h)) for _, h in results], dtype=torch.int32)
        buf = host.to(dev)  # one copy: [utterances, 1 + width] token ids of this rank
        gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, gathered, dst=0)
    barrier()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, audio_sec], dtype=torch.float64, device=dev)
    if dist_on:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, total_audio = float(tmax[0]), float(tsum[1])
    else:
        total_audio = audio_sec

    out = None
    if rank == 0:
        out = {
            "metric": "audio-sec/s decoded (node), Conformer-L beam=10", "value": round(total_audio / dt, 2),
            "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / max(args.steps, 1), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Conformer-L enc-dec ({args.attention}, 12+6 layers, d=512, V=5000) + "
                                   "S2STransformerBeamSearcher beam=10 + CTC 0.4"
                                   + (" + TransformerLM 12x768 scorer 0.6" if args.lm else "")
                                   + "; 16 kHz 0.1*randn audio, durations "
                                   "U(5,30) s, duration-sorted batches; decode steps = round(4 tok/s * seconds)",
                       "batch": args.batch, "utterances_per_gpu": args.steps * args.batch,
                       "audio_seconds_total": round(total_audio, 1), "weights": "random init, torch.manual_seed(0)",
                       "parallelism": f"replicas x{world}, utterance sharding, gather of token ids only",
                       "batches_in_flight_per_gpu": workers.n},
        }

    note(f"timed region done: {dt:.3f} s")
    # ---- p50 per-utterance latency (B = 1, 10 s), rank 0 only
    if rank == 0 and args.latency_runs > 0:
        w1 = (0.1 * torch.randn(1, 160000, generator=torch.Generator().manual_seed(5))).to(dev)
        l1 = torch.ones(1, device=dev)
        by_mode = {}
        lat_stream = torch.cuda.Stream(dev)  # (the legacy default stream cannot be captured into a graph)
        dec = asr.mods.decoder
        saved = (dec.overlap_ctc, dec.graph_mode)
        # two ways to run a single search: CTC scorer on a helper stream beside the decoder step, or the
        # decoding steps replayed from a captured hipGraph (device-side step counter); report the better one
        for mode, (ov, gm) in (("helper_stream", (3, 0)), ("hipgraph", (0, 1))):
            dec.overlap_ctc, dec.graph_mode = ov, gm
            lat = []
            with torch.cuda.stream(lat_stream):
                run_step(asr, w1, l1)
                for _ in range(args.latency_runs):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    run_step(asr, w1, l1)
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t)
            lat.sort()
            by_mode[mode] = round(1000.0 * lat[len(lat) // 2], 2)
        dec.overlap_ctc, dec.graph_mode = saved
        out["p50_latency_ms"] = min(by_mode.values())
        out["p50_latency_ms_by_mode"] = by_mode
        out["config"]["latency_case"] = "B=1, 10 s utterance, 40 decode steps"

    # ---- roofline of the dominant kernel: HIP events around every launch, same steps repeated
    if rank == 0 and not args.no_roofline:
        note("instrumented repetition (HIP events)")
        native.prof_reset()
        native.prof_enable(True)
        for w, l, _ in pool_dev:
            run_step(asr, w, l)
        torch.cuda.synchronize()
        native.prof_enable(False)
        rep = native.prof_report()
        native.prof_reset()
        total_ms = sum(v["ms"] for v in rep.values()) or 1.0
        name, top = max(rep.items(), key=lambda kv: kv[1]["ms"])
        mfma = name.startswith(MFMA_KERNELS)
        avg_ms = top["ms"] / top["count"]
        if mfma:
            ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4)}
        else:
            ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4)}
        traffic = None
        try:  # HBM bytes per launch from the PMC counters, collected in their own rocprofv3 --pmc passes (tools/run_pmc.sh)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = next((k for k in pmc if not k.startswith("_") and name.startswith(k)), None)
            if key:
                traffic = pmc[key]["bytes_per_launch"]
                roof["traffic_note"] = f"{pmc[key]['note']}; algorithmic {pmc[key]['algorithmic_bytes_per_launch']} B/launch"
            busy = {k: v["mfma_busy"] for k, v in pmc.get("_mfma_busy", {}).items()
                    if not k.startswith("_") and k.startswith(name)}
            if busy:  # SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x active cycles), from its own --pmc pass
                roof["mfma_busy_pmc"] = busy
        except Exception:
            pass
        roof.update({"traffic": traffic, "kernel": name, "launches": top["count"], "avg_launch_ms": round(avg_ms, 4),
                     "share_of_gpu_time": round(top["ms"] / total_ms, 3)})
        out["roofline"] = roof
        out["kernel_breakdown_ms"] = {k: round(v["ms"], 2) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (subprocess)")
        out["cpu_baseline"] = cpu_baseline_subprocess()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
