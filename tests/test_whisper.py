"""WhisperASR front-end (SURVEY 8f / BASELINE.json configs[4]): pad_or_trim + log_mel_spectrogram of
integrations/huggingface/whisper.py:276-350 on the HIP kernel, against (a) the reference's own function run here
(tests/golden/whisper_logmel.npz, oracle/make_golden.py --whisper-only) and (b) a plain torch fp32 restatement on
fresh inputs.  The reference has no unit test for Whisper: parity is pinned by running the reference."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "whisper_logmel.npz")


def torch_log_mel(audio, filters, n_fft=400, hop=160):
    stft = torch.stft(audio, n_fft, hop, window=torch.hann_window(n_fft), return_complex=True)
    mel = filters @ (stft[..., :-1].abs() ** 2)
    log_spec = torch.clamp(mel, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def test_slaney_filters_match_transformers():
    tf_audio = pytest.importorskip("transformers.audio_utils")
    from speechbrain_amd.integrations.huggingface.whisper import slaney_mel_filters

    for n_mels in (80, 128):
        ref = tf_audio.mel_filter_bank(num_frequency_bins=201, num_mel_filters=n_mels, min_frequency=0.0,
                                       max_frequency=8000.0, sampling_rate=16000, norm="slaney", mel_scale="slaney")
        assert float(np.abs(slaney_mel_filters(n_mels).numpy() - ref.astype(np.float32)).max()) <= 1e-7


@pytest.mark.parametrize("n_mels", [80, 128])
def test_log_mel_matches_reference_golden(backend, n_mels):
    nat, dev = backend
    from speechbrain_amd.integrations.huggingface.whisper import WhisperLogMel

    g = np.load(GOLD)
    fe = WhisperLogMel(n_mels=n_mels, n_samples=int(g["n_samples"]), mel_filters=g[f"filters{n_mels}"]).to(dev)
    wav = torch.from_numpy(g["wav"]).to(dev)
    out = fe(wav)
    ref = torch.from_numpy(g[f"mel{n_mels}"])
    assert out.shape == ref.shape
    assert float((out.cpu() - ref).abs().max()) <= 2e-4  # log10 of fp32 power sums; the floor is batch-global
    # trimming and the raw (un-padded) path
    long_wav = torch.cat([wav, wav], dim=1)
    assert torch.equal(fe.pad_or_trim(long_wav).cpu(), fe.pad_or_trim(wav.repeat(1, 2)).cpu())
    assert fe.pad_or_trim(long_wav).shape[1] == int(g["n_samples"])


def test_log_mel_vs_torch_restatement(backend):
    """Fresh seeded input incl. silence (the 1e-10 clamp) and a loud utterance that sets the batch-wide floor."""
    nat, dev = backend
    from speechbrain_amd.integrations.huggingface.whisper import WhisperLogMel

    fe = WhisperLogMel(n_mels=80, n_samples=16000).to(dev)
    g = torch.Generator().manual_seed(5)
    wav = torch.stack([0.5 * torch.randn(16000, generator=g), 1e-3 * torch.randn(16000, generator=g), torch.zeros(16000)])
    out = fe(wav.to(dev)).cpu()
    ref = torch_log_mel(wav, fe._mel_filters.cpu())
    assert out.shape == (3, 80, 100)
    assert float((out - ref).abs().max()) <= 2e-4
    assert float(out[2].max()) == pytest.approx(float(ref[2].max()), abs=1e-6)  # silence sits exactly on the floor


# ------------------------------------------------------------------------------------------- encoder / decoder
MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "whisper_tiny")
MODEL_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "whisper_model.npz")


def test_whisper_encoder_matches_reference_golden(backend):
    """forward_encoder (conv1/conv2 as strided-window GEMMs, positions, pre-norm layers with plain MHA on the flash
    kernel, final norm) against the REFERENCE wrapper's outputs for the tiny random model under tests/golden/
    whisper_tiny (oracle/make_golden.py:golden_whisper_model).  Tolerance 5e-5 absolute on activations of unit
    scale (|x| up to 3.7); the mel feeding it is the kernel's own (2e-4, its own test)."""
    nat, dev = backend
    from speechbrain_amd.integrations.huggingface.whisper import Whisper

    g = np.load(MODEL_GOLD)
    w = Whisper(MODEL_DIR, encoder_only=True).to(dev).eval()
    assert w.tokenizer is None and w.model.decoder is None
    mel_ref = torch.from_numpy(g["mel"]).to(dev)
    enc = w.forward_encoder(mel_ref)
    assert enc.shape == g["enc"].shape
    assert float((enc.cpu() - torch.from_numpy(g["enc"])).abs().max()) <= 5e-5
    # all hidden states, stacked as the reference stacks them
    w.output_all_hiddens = True
    allh = w.forward_encoder(mel_ref)
    assert allh.shape == g["enc_all"].shape
    assert float((allh.cpu() - torch.from_numpy(g["enc_all"])).abs().max()) <= 5e-5
    w.output_all_hiddens = False
    # from the waveform: the wrapper's own mel (1-second chunk of this tiny model) -> encoder
    wav = torch.from_numpy(g["wav"]).to(dev)
    mel = w.log_mel_spectrogram(w.pad_or_trim(wav, 16000))
    assert float((mel.cpu() - torch.from_numpy(g["mel"])).abs().max()) <= 2e-4
    enc2 = w.forward_encoder(mel)
    assert float((enc2.cpu() - torch.from_numpy(g["enc"])).abs().max()) <= 1e-3
    with pytest.raises(ValueError):  # _get_mel pads to 30 s like the reference: 3000 frames do not fit this model
        w(wav)


class _StubTokenizer:
    """The handful of tokenizer calls the Whisper wrapper / searcher make, over the tiny model's 100-token vocabulary
    (the same ids the golden was generated with: oracle/make_golden.py WHISPER_IDS)."""

    IDS = {"<|endoftext|>": 2, "<|startoftranscript|>": 3, "<|en|>": 4, "<|fr|>": 5, "<|transcribe|>": 10,
           "<|translate|>": 11, "<|startoflm|>": 12, "<|startofprev|>": 13, "<|nospeech|>": 14, "<|notimestamps|>": 15}
    prefix_tokens = [3, 4, 10, 15]

    def convert_tokens_to_ids(self, token):
        return self.IDS[token]

    def encode(self, text, add_special_tokens=False):
        return {" ": [16]}[text]


def _full_whisper(dev):
    from speechbrain_amd.integrations.huggingface.whisper import Whisper

    w = Whisper(MODEL_DIR).to(dev).eval()
    assert w.tokenizer is None  # the fixture directory holds no tokenizer files
    w.tokenizer = _StubTokenizer()
    w._non_speech = (20, 21, 22, 40)
    return w


def test_whisper_decoder_logits_match_reference_golden(backend):
    """forward_decoder (teacher-forced prefix through the KV-cached decoder step: learned positions, unscaled tied
    embedding, k_proj without bias, GELU) against the reference wrapper's logits.  Tolerance 1e-4 on logits up to 2.7."""
    nat, dev = backend
    g = np.load(MODEL_GOLD)
    w = _full_whisper(dev)
    enc = torch.from_numpy(g["enc"]).to(dev)
    logits, attn, cache = w.forward_decoder(enc, torch.from_numpy(g["tokens"]).to(dev))
    assert attn is None and cache is None
    assert logits.shape == g["logits"].shape
    assert float((logits.cpu() - torch.from_numpy(g["logits"])).abs().max()) <= 1e-4
    # forward(): mel -> encoder -> decoder in one call, on a 30-second model this is the reference's entry point;
    # here it must refuse the 3000-frame mel just like the encoder does
    with pytest.raises(ValueError):
        w(torch.from_numpy(g["wav"]).to(dev), torch.from_numpy(g["tokens"]).to(dev))


def test_whisper_greedy_searcher_matches_reference_golden(backend):
    """S2SWhisperGreedySearcher: initial tokens with a per-utterance language token, suppress lists ("-1" + specials,
    blank and EOS at the first sampled step), EOS latch, stop once every utterance has ended, no_speech_probs --
    token ids equal the reference's (its own top-1 / top-2 margin is >= 0.1), per-step log-probs within 2e-4."""
    nat, dev = backend
    from speechbrain_amd.decoders import S2SWhisperGreedySearcher

    g = np.load(MODEL_GOLD)
    w = _full_whisper(dev)
    s = S2SWhisperGreedySearcher(model=w, min_decode_ratio=0.0, max_decode_ratio=1.0)
    assert list(s.initial_tokens) == g["initial_tokens"].tolist()
    assert list(s.get_tokens_to_suppress) == g["suppress"].tolist()
    s.set_lang_tokens(torch.tensor([4, 5, 6]))
    enc = torch.from_numpy(g["enc"]).to(dev)
    hyps, lens, scores, _ = s(enc, torch.ones(3))
    ref_h = [[int(t) for t in row if t >= 0] for row in g["greedy_hyps"]]
    assert hyps == ref_h
    assert torch.allclose(lens.reshape(-1), torch.from_numpy(g["greedy_lens"]).reshape(-1).float(), atol=1e-6)
    assert scores.shape == g["greedy_scores"].shape
    assert float((scores - torch.from_numpy(g["greedy_scores"])).abs().max()) <= 2e-4
    assert np.abs(np.array(s.no_speech_probs) - g["greedy_no_speech"]).max() <= 1e-5
    # without masks the first sampled token may be anything: the masks are what keeps blank / EOS / specials out
    free = S2SWhisperGreedySearcher(model=w, suppress_blank=False, suppress_tokens=[])
    h2, _, sc2, _ = free(enc, torch.ones(3))
    assert sc2.shape[2] >= 1 and all(len(h) <= sc2.shape[2] for h in h2)


def test_whisper_asr_interface(backend):
    """inference.ASR.WhisperASR: encode_batch / transcribe_batch wiring (mods.whisper + mods.decoder + the tokenizer's
    decode).  The 30-second chunk of _get_mel needs a 1500-position encoder; the tiny fixture has 50, so the mel step is
    replaced by the 1-second one here and the pieces are exercised end to end."""
    nat, dev = backend
    from speechbrain_amd.decoders import S2SWhisperGreedySearcher
    from speechbrain_amd.inference.ASR import WhisperASR

    g = np.load(MODEL_GOLD)
    w = _full_whisper(dev)
    w.tokenizer.decode = lambda toks, skip_special_tokens=True: " " + " ".join(f"t{t}" for t in toks) + " "
    w._get_mel = lambda wav: w.log_mel_spectrogram(w.pad_or_trim(wav, 16000))
    searcher = S2SWhisperGreedySearcher(model=w)
    searcher.set_lang_tokens(torch.tensor([4, 5, 6]))
    asr = WhisperASR(modules={"whisper": w, "decoder": searcher},
                     hparams={"language": "en", "sample_rate": 16000, "whisper": w, "normalized_transcripts": False},
                     run_opts={"device": str(dev)})
    words, tokens = asr.transcribe_batch(torch.from_numpy(g["wav"]), torch.ones(3))
    ref_h = [[int(t) for t in row if t >= 0] for row in g["greedy_hyps"]]
    assert tokens == ref_h
    assert words[0] == " ".join(f"t{t}" for t in ref_h[0])
    enc = asr.encode_batch(torch.from_numpy(g["wav"]), torch.ones(3))
    assert float((enc.cpu() - torch.from_numpy(g["enc"])).abs().max()) <= 1e-3


def test_whisper_asr_run_opts_precision_reaches_the_fp8_pipeline(backend):
    """BASELINE configs[4] "via speechbrain.inference, fp8 MFMA": WhisperASR(run_opts={"precision": "fp8"}) runs its encoder
    inside native.precision_scope("fp8") (the reference builds the inference context of every forward from run_opts,
    inference/interfaces.py:295-298) -- the layer's four contractions take native.gemm_nt_fp8a (encode_batch here; transcribe_batch
    and transcribe_file_streaming go through the same WhisperASR._encode_mel); the default interface stays on fp32 and takes none; fp16 / bf16 are
    accepted, anything else is refused by name."""
    nat, dev = backend
    from speechbrain_amd.inference.ASR import WhisperASR
    from speechbrain_amd.integrations.huggingface.whisper import Whisper

    cfg = dict(num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256,
               max_source_positions=160, decoder_layers=1, decoder_attention_heads=2, decoder_ffn_dim=256,
               vocab_size=100, max_target_positions=16)
    w = Whisper.from_config(cfg, seed=4).to(dev).eval()
    w._get_mel = lambda wav: w.log_mel_spectrogram(w.pad_or_trim(wav, 51200))  # 320 frames -> 160 encoder positions
    hp = {"language": "en", "sample_rate": 16000, "whisper": w, "normalized_transcripts": False}
    wav = 0.1 * torch.randn(2, 51200, generator=torch.Generator().manual_seed(9))
    calls = {"fp8a": 0}
    g0 = nat.gemm_nt_fp8a

    def counted(*a, **k):
        calls["fp8a"] += 1
        return g0(*a, **k)

    nat.gemm_nt_fp8a = counted
    try:
        with torch.no_grad():
            a32 = WhisperASR(modules={"whisper": w, "decoder": torch.nn.Identity()}, hparams=hp, run_opts={"device": str(dev)})
            ref = a32.encode_batch(wav, torch.ones(2))
            assert calls["fp8a"] == 0 and a32.eval_precision == "fp32"
            a8 = WhisperASR(modules={"whisper": w, "decoder": torch.nn.Identity()}, hparams=hp,
                            run_opts={"device": str(dev), "precision": "fp8"})
            got = a8.encode_batch(wav, torch.ones(2))
            assert calls["fp8a"] == 4 * 2  # the four contractions of each of the two layers
            assert nat.precision() == "fp32"  # (the scope ends with the forward)
    finally:
        nat.gemm_nt_fp8a = g0
    rel = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert 0.0 < rel <= 8e-2, rel  # e4m3 operands: the tolerance of test_whisper_encoder_fp8_activation_pipeline
    for p in ("bf16", "fp16"):
        assert WhisperASR(modules={"whisper": w, "decoder": torch.nn.Identity()}, hparams=hp,
                          run_opts={"device": str(dev), "eval_precision": p}).eval_precision == p
    with pytest.raises(NotImplementedError):
        WhisperASR(modules={"whisper": w, "decoder": torch.nn.Identity()}, hparams=hp, run_opts={"device": str(dev), "precision": "int4"})


def test_whisper_beam_searcher_matches_reference_golden(backend):
    """S2SWhisperBeamSearcher (seq2seq.py:1937-2206) on the device search: prompt priming of every hypothesis' KV cache,
    per-utterance language tokens, suppression masks, log_softmax / temperature, eos rules, return_topk -- token ids
    equal the reference's, scores within 2e-4, no_speech_probs within 1e-5; temperatures 1 and 0.8, a token prompt,
    a minimum length without length normalisation, the top-3 lists."""
    nat, dev = backend
    from speechbrain_amd.decoders import S2SWhisperBeamSearcher

    g = np.load(MODEL_GOLD)
    w = _full_whisper(dev)
    enc = torch.from_numpy(g["enc"]).to(dev)
    base = dict(beam_size=4, min_decode_ratio=0.0, max_decode_ratio=1.0, using_eos_threshold=False, length_normalization=True)
    for tag, kw in (("t10", dict(temperature=1.0)), ("t08", dict(temperature=0.8)),
                    ("prompt", dict(temperature=1.0, prompt=[30, 31, 32])),
                    ("min6", dict(temperature=1.0, min_decode_ratio=0.12, length_normalization=False))):
        s = S2SWhisperBeamSearcher(module=[w], **{**base, **kw})
        assert list(s.initial_tokens) == g[f"beam_{tag}_init"].tolist()
        s.set_lang_tokens(torch.tensor([4, 5, 6]).repeat_interleave(4))  # one per hypothesis, as the reference takes them
        hyps, lens, scores, _ = s(enc, torch.ones(3))
        assert hyps == [[int(t) for t in row if t >= 0] for row in g[f"beam_{tag}_hyps"]], tag
        assert float((scores.cpu() - torch.from_numpy(g[f"beam_{tag}_scores"])).abs().max()) <= 2e-4, tag
        assert float((lens.cpu().float() - torch.from_numpy(g[f"beam_{tag}_lens"]).float()).abs().max()) <= 1e-6, tag
        assert np.abs(np.array(s.no_speech_probs) - g[f"beam_{tag}_no_speech"][::4]).max() <= 1e-5, tag
        s.set_lang_tokens(torch.tensor([4, 5, 6]))  # one per utterance gives the same search
        assert s(enc, torch.ones(3))[0] == hyps
    # forward_group: a token prompt primes every hypothesis' cache at its own decoder positions, so the batches of a group keep
    # their own searches (two copies of the batch give the golden hypotheses twice)
    s = S2SWhisperBeamSearcher(module=[w], **{**base, "temperature": 1.0})
    s.set_lang_tokens(torch.tensor([4, 5, 6]))
    both = s.forward_group([(enc, torch.ones(3)), (enc, torch.ones(3))])
    assert len(both) == 2 and both[0][0] == both[1][0] == [[int(t) for t in row if t >= 0] for row in g["beam_t10_hyps"]]
    s = S2SWhisperBeamSearcher(module=[w], **{**base, "temperature": 1.0, "return_topk": True, "topk": 3})
    s.set_lang_tokens(torch.tensor([4, 5, 6]))
    k_hyps, k_lens, k_scores, _ = s(enc, torch.ones(3))
    assert torch.equal(k_hyps.cpu(), torch.from_numpy(g["beam_top3_hyps"]))
    assert float((k_scores.cpu() - torch.from_numpy(g["beam_top3_scores"])).abs().max()) <= 2e-4


def test_whisper_language_identification_and_file_transcription(backend, tmp_path):
    """Whisper.detect_language (whisper.py:617-665) against the reference's tokens / probabilities, and
    WhisperASR.detect_language_batch / transcribe_file (inference/ASR.py:475-865): segments of the file, running prompt,
    lang_id task, the no-speech skip rule."""
    nat, dev = backend
    import struct
    import wave

    from speechbrain_amd.decoders import S2SWhisperBeamSearcher
    from speechbrain_amd.inference.ASR import ASRWhisperSegment, WhisperASR

    g = np.load(MODEL_GOLD)
    w = _full_whisper(dev)
    w.tokenizer.language, w.tokenizer.bos_token = "en", "<|startoftranscript|>"
    w._lang_tokens, w._lang_codes = (4, 5, 6, 7), ("en", "fr", "de", "es")
    toks, probs = w.detect_language(torch.from_numpy(g["mel"]).to(dev))
    assert toks.cpu().tolist() == g["lang_tokens"].tolist()
    got = np.array([[p[c] for c in w._lang_codes] for p in probs])
    assert np.abs(got - g["lang_probs"]).max() <= 1e-5
    # the interface: 1-second "chunks" for the 50-position toy encoder
    w.tokenizer.decode = lambda t, skip_special_tokens=True: " " + " ".join(f"t{x}" for x in t) + " "
    w.tokenizer.encode = lambda text, add_special_tokens=False: {" ": [16]}.get(text, [30, 31])
    w._get_mel = lambda wav: w.log_mel_spectrogram(w.pad_or_trim(wav, 16000))
    searcher = S2SWhisperBeamSearcher(module=[w], beam_size=4, using_eos_threshold=False)
    asr = WhisperASR(modules={"whisper": w, "decoder": searcher},
                     hparams={"language": None, "sample_rate": 16000, "whisper": w}, run_opts={"device": str(dev)})
    w.language = None  # language unknown: every segment is identified first
    lt, lp = asr.detect_language_batch(torch.from_numpy(g["wav"]))
    assert lt.cpu().tolist() == g["lang_tokens"].tolist()
    path = str(tmp_path / "two_and_a_half_seconds.wav")
    pcm = (np.concatenate([g["wav"][0], g["wav"][1], g["wav"][2][:8000]]) * 32767 / 2).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
    segs = asr.transcribe_file(path, chunk_size=1, no_speech_threshold=None)
    assert [type(s) for s in segs] == [ASRWhisperSegment] * 3 and [(s.start, s.end) for s in segs] == [(0, 1), (1, 2), (2, 3)]
    assert all(s.lang_id == "fr" and s.tokens and s.words == " ".join(f"t{x}" for x in s.tokens) for s in segs)
    assert segs[0].prompt == [] and all(0.0 <= s.no_speech_prob <= 1.0 for s in segs)
    ids = asr.transcribe_file(path, task="lang_id", chunk_size=1)
    assert [s.lang_id for s in ids] == ["fr"] * 3 and ids[0].words is None
    skipped = asr.transcribe_file(path, chunk_size=1, no_speech_threshold=0.0, logprob_threshold=None)
    assert all(s.words == "" and s.tokens == [] for s in skipped)  # every segment is above a zero no-speech threshold
    prompted = asr.transcribe_file(path, chunk_size=1, initial_prompt="hello", no_speech_threshold=None)
    assert prompted[0].prompt == [30, 31]


def test_whisper_encoder_bf16_activation_pipeline(backend):
    """precision "bf16": the encoder keeps the operands of its contractions in bf16 in memory (LayerNorm / attention /
    GELU epilogue write bf16, native.gemm_nt_bf16a reads them by LDS-DMA).  Same roundings as the kernels that read fp32
    activations and round on load (native.BF16_ACTIVATIONS = False): both within the bf16 tolerance of the fp32
    encoder, and of each other (they differ by summation order only, so mostly bit-identical roundings)."""
    nat, dev = backend
    from speechbrain_amd.integrations.huggingface.whisper import Whisper

    cfg = dict(num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256,
               max_source_positions=40, decoder_layers=0, decoder_attention_heads=2, decoder_ffn_dim=256,
               vocab_size=100, max_target_positions=16)
    w = Whisper.from_config(cfg, encoder_only=True, seed=4).to(dev).eval()
    mel = torch.randn(2, 80, 80, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        ref = w.model.encoder(mel)
        with nat.precision_scope("bf16"):
            got_a = w.model.encoder(mel)
            nat.BF16_ACTIVATIONS = False
            try:
                got_f = w.model.encoder(mel)
            finally:
                nat.BF16_ACTIVATIONS = True
    assert got_a.dtype == torch.float32 and got_a.shape == ref.shape
    rms = float(ref.pow(2).mean().sqrt())
    for got in (got_a, got_f):
        err = (got - ref).float()
        assert float(err.abs().max()) <= 0.15 and float(err.pow(2).mean().sqrt()) <= 1e-2 * rms
    assert float((got_a - got_f).pow(2).mean().sqrt()) <= 3e-3 * rms


def test_whisper_encoder_fp8_activation_pipeline(backend):
    """precision "fp8" (BASELINE configs[4]): LayerNorm writes e4m3 rows with one scale per row, the weights carry one
    scale per output channel, the four contractions of a layer run on the 2 x-rate fp8 matrix instruction
    (native.gemm_nt_fp8a; hidden layer handed over as e4m3; attention on bf16 rows, its context quantised row by row).  Against
    the fp32 encoder: e4m3 keeps 3 significand bits -- absolute 1.0, relative RMS 8 % (the tolerance of the round-3 fp8
    path with fp32 activations and per-tensor scales, which must hold too, and be no tighter than the new one)."""
    nat, dev = backend
    from speechbrain_amd.integrations.huggingface.whisper import Whisper

    cfg = dict(num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256,
               max_source_positions=160, decoder_layers=0, decoder_attention_heads=2, decoder_ffn_dim=256,
               vocab_size=100, max_target_positions=16)
    w = Whisper.from_config(cfg, encoder_only=True, seed=4).to(dev).eval()
    mel = torch.randn(2, 80, 320, generator=torch.Generator().manual_seed(5)).to(dev)  # 2 x 160 frames = 320 rows
    calls = {"fp8a": 0}
    g0 = nat.gemm_nt_fp8a

    def counted(*a, **k):
        calls["fp8a"] += 1
        return g0(*a, **k)

    nat.gemm_nt_fp8a = counted
    try:
        with torch.no_grad():
            ref = w.model.encoder(mel)
            with nat.precision_scope("fp8"):
                got_a = w.model.encoder(mel)
                nat.FP8_ACTIVATIONS = False
                try:
                    got_t = w.model.encoder(mel)
                finally:
                    nat.FP8_ACTIVATIONS = True
    finally:
        nat.gemm_nt_fp8a = g0
    assert calls["fp8a"] == 4 * 2  # the four contractions of a layer on the fp8 instruction
    rms = float(ref.pow(2).mean().sqrt())
    errs = []
    for got in (got_a, got_t):
        err = (got - ref).float()
        errs.append(float(err.pow(2).mean().sqrt()) / rms)
        assert float(err.abs().max()) <= 1.0 and errs[-1] <= 8e-2, errs
    assert errs[0] <= 1.25 * errs[1] + 1e-3, errs  # per-row / per-channel scales: no worse than per-tensor scaling
