"""Full-size parity on the MI355X (gpu only): Conformer-S / Conformer-L shapes of BASELINE.json
against the oracle on the same seeded weights, plus size-independent properties at the bench shape."""
import math

import pytest
import torch

from oracle import sb_oracle as O

pytestmark = pytest.mark.gpu


def _asr(size, **kw):
    from speechbrain_amd import native
    from speechbrain_amd.inference.builders import build_asr

    import emu_utils

    emu_utils.detach()
    native.load()
    return build_asr(size, device="cuda:0", **kw)


def _oracle_cfg(size):
    from speechbrain_amd.inference.builders import SIZES

    c = SIZES[size]
    fc = O.FbankCfg(n_fft=c["n_fft"], n_mels=80, win_length_ms=c["win_length"])
    mc = O.ModelCfg(d_model=c["d_model"], nhead=c["nhead"], num_encoder_layers=c["n_enc"],
                    num_decoder_layers=c["n_dec"], d_ffn=c["d_ffn"], vocab=5000)
    return fc, mc


@pytest.mark.parametrize("size,B,sec", [("S", 4, 10.0), ("S", 32, 10.0), ("L", 2, 6.0), ("L", 2, 22.0), ("L", 12, 10.0),
                                        ("L", 32, 8.0)])
def test_encoder_vs_oracle(size, B, sec):
    """configs[1] (Conformer-S, 32 x 10 s -- BASELINE.json's exact shape -- and a 4-utterance cut of it) and
    Conformer-L encoders (6 s, and 22 s = T' 551: the headline's utterance lengths): Fbank within 1e-3 dB,
    encoder output within 2e-4 absolute (fp32; oracle self-noise is 2e-6).

    The encoder's dominant kernel inside an oracle comparison: L-12-10 s has M = 3 012 rows, so
    the contractions with N = 2 048 / 1 536 / 1 024 take the split-operand route (round 4: sbk_gemm_nt_x3p, both
    operands pre-split -- LayerNorm writes the panel operand, the feed-forward hidden layer is handed over as a panel
    image; sbk_gemm_nt_f32x3 with SBK_X3P=0) and those with N = 512 the fp32-MFMA kernels -- both meet in every layer;
    L-32-8 s has M = 6 432 rows: EVERY contraction of the layer takes it.  The routes are asserted, not assumed."""
    from speechbrain_amd import native
    from speechbrain_amd.inference.builders import flat_state_dict

    asr = _asr(size)
    n = int(sec * 16000)
    if size == "L" and B >= 12:
        M = B * (((1 + n // 160 - 1) // 2 + 1 - 1) // 2 + 1)
        d = 512
        routes = {N: native.f32x3_ok(M, K, torch.empty(N, K)) for N, K in ((2048, d), (3 * d, d), (2 * d, d), (d, 2048), (d, d))}
        assert native.F32X3 and routes[2048] and routes[3 * d] and routes[2 * d], routes
        assert routes[d] == (B == 32), routes
        if native.X3P:  # (the default: the same shapes on the both-operands-pre-split kernel)
            proutes = {N: native.x3p_ok(M, K, torch.empty(N, K)) for N, K in ((2048, d), (3 * d, d), (2 * d, d), (d, 2048), (d, d))}
            assert proutes[2048] and proutes[3 * d] and proutes[2 * d] and proutes[d] == (B == 32), proutes
    wav = 0.1 * torch.randn(B, n, generator=torch.Generator().manual_seed(1234))
    lens = torch.linspace(0.55, 1.0, B)
    for i in range(B):
        wav[i, int(lens[i] * n):] = 0
    fc, mc = _oracle_cfg(size)
    sd = flat_state_dict(asr)
    feats = asr.mods.encoder["compute_features"](wav.cuda())
    assert float((feats.cpu() - O.fbank(wav, fc)).abs().max()) <= 1e-3
    enc = asr.encode_batch(wav, lens).cpu()
    ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    assert enc.shape == ref.shape
    assert float((enc - ref).abs().max()) <= 2e-4


def test_rope_conformer_l_encoder_vs_oracle():
    """Conformer-L with attention_type=RoPEMHA (the in-tree conformer_large.yaml:158): encoder output
    within 2e-4 of the oracle, with padding."""
    import dataclasses

    from speechbrain_amd.inference.builders import flat_state_dict

    asr = _asr("L", attention_type="RoPEMHA")
    n = 6 * 16000
    wav = 0.1 * torch.randn(2, n, generator=torch.Generator().manual_seed(1234))
    lens = torch.tensor([1.0, 0.55])
    wav[1, int(0.55 * n):] = 0
    fc, mc = _oracle_cfg("L")
    mc = dataclasses.replace(mc, attention_type="RoPEMHA")
    sd = flat_state_dict(asr)
    enc = asr.encode_batch(wav, lens).cpu()
    ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    assert float((enc - ref).abs().max()) <= 2e-4


def test_conformer_l_decoder_logprobs_and_search_vs_oracle():
    """Conformer-L, beam 10 + CTC 0.4: first-step log-probs within 1e-4, then (with peaked output
    heads so that fp32 noise cannot flip a near-tie, SURVEY 7 hard-part 1) bit-exact token ids and
    scores within 1e-3 against the oracle's full-prefix / Python-loop CTC search."""
    from speechbrain_amd import native
    from speechbrain_amd.inference.builders import flat_state_dict

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    fc, mc = _oracle_cfg("L")
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
        asr.mods.ctc_lin.w.weight.mul_(8.0)
    n = 3 * 16000
    wav = 0.1 * torch.randn(2, n, generator=torch.Generator().manual_seed(4))
    lens = torch.tensor([1.0, 0.7])
    wav[1, int(0.7 * n):] = 0
    sd = flat_state_dict(asr)
    enc_ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    T = enc_ref.shape[1]
    enc_lens = torch.round(T * lens).int()
    # decoder + seq_lin + log_softmax on a fixed prefix
    tgt = torch.tensor([[1, 17, 4, 250], [1, 9, 9, 3000]])
    h = native.DecoderHandle(asr.mods.transformer, asr.mods.seq_lin)
    pred = native.decoder_prefix(h, tgt.int().cuda(), enc_ref.cuda(), enc_lens.cuda())
    lp = native.log_softmax(native.gemm_nt(pred, asr.mods.seq_lin.w.weight, asr.mods.seq_lin.w.bias))
    dec_ref = O.decode(tgt, enc_ref, enc_lens, sd, mc, "Transformer.")
    lp_ref = torch.log_softmax(torch.nn.functional.linear(dec_ref, sd["seq_lin.w.weight"], sd["seq_lin.w.bias"]), -1)
    assert float((lp.cpu() - lp_ref).abs().max()) <= 1e-3  # heads scaled x8: 1e-4 on unscaled logits
    # search from the oracle's encoder output, 12 steps
    asr.mods.decoder.max_decode_ratio = 12.5 / T
    hyps, lens_o, scores, _ = asr.mods.decoder(enc_ref.cuda(), lens.cuda())
    tr = O.SearchTrace()
    hyps_ref, _, scores_ref, _ = O.beam_search(enc_ref, lens, sd, mc,
                                               O.SearchCfg(beam=10, ctc_weight=0.4, max_decode_ratio=12.5 / T), trace=tr)
    assert hyps == hyps_ref
    assert float((scores.cpu() - scores_ref).abs().max()) <= 1e-3
    # the same with the step's projections on sbk_gemm_nt_x3r (round 4: the route of every step with ~200 hypothesis rows
    # or more -- the bench's grouped searches; this search has 20 rows, so the row threshold is lowered, knob 42): the
    # kernel the headline's decode loop spends its time in, inside the oracle comparison at Conformer-L size
    lib = native.load()
    lib.sbk_prof_set_knob(42, 1)
    lib.sbk_prof_set_knob(47, 0)  # (the 2-row prefix above ran as the persistent few-row step, csrc/decoder_persist.hip; here the x3r route is wanted)
    try:
        assert h.layers[0].sa_in_wp and h.layers[0].ff2_wp and h.W.seq_wp
        pred3 = native.decoder_prefix(h, tgt.int().cuda(), enc_ref.cuda(), enc_lens.cuda())
        lp3 = native.log_softmax(native.gemm_nt(pred3, asr.mods.seq_lin.w.weight, asr.mods.seq_lin.w.bias))
        assert float((lp3.cpu() - lp_ref).abs().max()) <= 1e-3
        assert not torch.equal(pred3, pred)  # (another kernel did run)
        hyps3, _, scores3, _ = asr.mods.decoder(enc_ref.cuda(), lens.cuda())
        assert hyps3 == hyps_ref
        assert float((scores3.cpu() - scores_ref).abs().max()) <= 1e-3
    finally:
        lib.sbk_prof_set_knob(42, 192)
        lib.sbk_prof_set_knob(47, 1)


def test_ring_cross_attention_conformer_l_geometry_default_knobs_vs_oracle():
    """csrc/decoder.hip cross_attn_ring_kernel at the geometry and under the knobs it SHIPS with, inside an
    oracle comparison on the GPU.  Conformer-L decoder (d 512, 8 heads of 64, 6 layers), 16 utterances = 128 (utterance, head)
    pairs -- the default routing's threshold (knob 4 = 7) --, memories of T' = 420 frames with lengths 0.45 .. 1 of that, so the
    default run rule cuts each memory into 4 runs of 112 frames (partials, cross_merge; the short utterances leave EMPTY later
    runs, the last run of the longest holds 84 frames = 5 tiles + a 4-frame tile), beam 10 (a 16-column MFMA tile with six
    padding columns), no CTC (the oracle's Python-loop scorer would take minutes).  Then 32 utterances (256 pairs: the other
    side of the run rule, ns = 4 again but two waves' worth of workgroups per run).  Token ids exact (heads x8), scores 1e-3,
    teacher-forced decoder outputs 2e-4 -- and the profiler must show that the ring kernel and the merge are what ran."""
    from speechbrain_amd import native

    asr = _asr("L", beam_size=10, ctc_weight=0.0)
    _, mc = _oracle_cfg("L")
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
    from speechbrain_amd.inference.builders import flat_state_dict

    sd = flat_state_dict(asr)
    T = 420
    for B, steps in ((16, 10), (32, 4)):
        g = torch.Generator().manual_seed(100 + B)
        enc = torch.randn(B, T, 512, generator=g)
        lens = torch.linspace(0.45, 1.0, B)
        enc_lens = torch.round(T * lens).int()
        ratio = (steps + 0.5) / T
        asr.mods.decoder.max_decode_ratio = ratio
        native.prof_reset()
        native.prof_enable(True)
        try:
            hyps, _, scores, _ = asr.mods.decoder(enc.cuda(), lens.cuda())
        finally:
            native.prof_enable(False)
        rep = native.prof_report()
        native.prof_reset()
        assert rep.get("cross_attn_ring", {}).get("count", 0) >= 6 * steps and "cross_attn_step" not in rep, sorted(rep)
        assert rep.get("cross_merge", {}).get("count", 0) == rep["cross_attn_ring"]["count"], sorted(rep)  # (4 runs per memory: merged)
        hyps_ref, _, scores_ref, _ = O.beam_search(enc, lens, sd, mc, O.SearchCfg(beam=10, ctc_weight=0.0, max_decode_ratio=ratio))
        assert hyps == hyps_ref, B
        L = min(len(h) for h in hyps)
        assert L >= 3  # (a degenerate hypothesis would make the comparison vacuous)
        assert float((scores.cpu() - scores_ref).abs().max()) <= 1e-3
        if B == 16:  # teacher-forced decoder outputs on the search's own best hypotheses (one hypothesis row per memory: 16 x 8 pairs)
            tgt = torch.tensor([[1] + list(h[: L - 1]) for h in hyps])
            h = native.DecoderHandle(asr.mods.transformer, asr.mods.seq_lin)
            lib = native.load()
            lib.sbk_prof_set_knob(47, 0)  # (16 rows would otherwise run as the persistent few-row step, which has its own attention)
            native.prof_enable(True)
            try:
                pred = native.decoder_prefix(h, tgt.int().cuda(), enc.cuda(), enc_lens.cuda())
            finally:
                native.prof_enable(False)
                lib.sbk_prof_set_knob(47, 1)
            rep = native.prof_report()
            native.prof_reset()
            assert rep.get("cross_attn_ring", {}).get("count", 0) > 0 and rep.get("cross_merge", {}).get("count", 0) > 0, sorted(rep)
            ref = O.decode(tgt, enc, enc_lens, sd, mc, "Transformer.")
            assert float((pred.cpu() - ref).abs().max()) <= 2e-4


def test_recipe_lm_scorer_search_vs_oracle():
    """a20 at recipe size: Conformer-L + TransformerLM (12 x 768, 12 heads, d_ffn 3072, GELU, post-norm,
    LM temperature 1.15, lm_weight 0.6) + CTC 0.4 (conformer_large.yaml:166-223 ``test_search`` scorers,
    beam 10): LM logits within 2e-3 on a fixed prefix, then bit-exact token ids vs the oracle."""
    from speechbrain_amd.decoders import CTCScorer, S2STransformerBeamSearcher, ScorerBuilder, TransformerLMScorer
    from speechbrain_amd.inference.builders import flat_state_dict
    from speechbrain_amd.lobes.models.transformer.TransformerLM import TransformerLM

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    fc, mc = _oracle_cfg("L")
    torch.manual_seed(11)
    lm = TransformerLM(vocab=5000, d_model=768, nhead=12, num_encoder_layers=12, num_decoder_layers=0, d_ffn=3072,
                       dropout=0.0, activation=torch.nn.GELU, normalize_before=False).cuda().eval()
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
        asr.mods.ctc_lin.w.weight.mul_(8.0)
        lm.output_proj.layers[2].w.weight.mul_(4.0)
    lcfg = O.LMCfg(vocab=5000, d_model=768, nhead=12, num_encoder_layers=12, d_ffn=3072)
    sd = flat_state_dict(asr)
    sd.update({"LM." + k: v.detach().cpu() for k, v in lm.state_dict().items()})
    toks = torch.randint(1, 5000, (3, 14), generator=torch.Generator().manual_seed(5))
    toks[1, 4] = 0  # a masked key
    ref = O.lm_forward(toks, sd, lcfg, "LM.")
    got = lm(toks.cuda()).cpu()
    assert float((got - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()) / 10)
    n = 3 * 16000
    wav = 0.1 * torch.randn(2, n, generator=torch.Generator().manual_seed(4))
    lens = torch.tensor([1.0, 0.7])
    wav[1, int(0.7 * n):] = 0
    enc_ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    T = enc_ref.shape[1]
    scorer = ScorerBuilder(full_scorers=[TransformerLMScorer(language_model=lm, temperature=1.15),
                                         CTCScorer(ctc_fc=asr.mods.ctc_lin, blank_index=0, eos_index=2)],
                           weights={"transformerlm": 0.6, "ctc": 0.4})
    bs = S2STransformerBeamSearcher(modules=[asr.mods.transformer, asr.mods.seq_lin], bos_index=1, eos_index=2,
                                    min_decode_ratio=0.0, max_decode_ratio=10.5 / T, beam_size=10,
                                    using_eos_threshold=False, length_normalization=True, temperature=1.15,
                                    scorer=scorer)
    hyps, _, scores, _ = bs(enc_ref.cuda(), lens.cuda())
    hyps_ref, _, scores_ref, _ = O.beam_search(
        enc_ref, lens, sd, mc, O.SearchCfg(beam=10, ctc_weight=0.4, max_decode_ratio=10.5 / T, temperature=1.15,
                                           lm_weight=0.6, lm_temperature=1.15), lm_cfg=lcfg)
    assert hyps == hyps_ref
    assert float((scores.cpu() - scores_ref).abs().max()) <= 2e-3
    hyps_nolm, _, _, _ = O.beam_search(enc_ref, lens, sd, mc, O.SearchCfg(beam=10, ctc_weight=0.4,
                                                                           max_decode_ratio=10.5 / T, temperature=1.15))
    assert hyps_nolm != hyps_ref  # the LM is not a no-op in this test


def test_recipe_test_search_beam66_vs_oracle():
    """The recipe's test_search at full width (conformer_large.yaml:241-251: beam 66, TransformerLM 0.6 +
    CTC 0.4, temperature 1.15) on Conformer-L: radix-select top-k, CTC tables in 5 tiles of 16 beams,
    LM K/V cache for 66 hypotheses -- bit-exact token ids and the top-3 list vs the oracle."""
    from speechbrain_amd.decoders import CTCScorer, S2STransformerBeamSearcher, ScorerBuilder, TransformerLMScorer
    from speechbrain_amd.inference.builders import flat_state_dict
    from speechbrain_amd.lobes.models.transformer.TransformerLM import TransformerLM

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    fc, mc = _oracle_cfg("L")
    torch.manual_seed(12)
    lm = TransformerLM(vocab=5000, d_model=768, nhead=12, num_encoder_layers=12, num_decoder_layers=0, d_ffn=3072,
                       dropout=0.0, activation=torch.nn.GELU, normalize_before=False).cuda().eval()
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
        asr.mods.ctc_lin.w.weight.mul_(8.0)
        lm.output_proj.layers[2].w.weight.mul_(4.0)
    lcfg = O.LMCfg(vocab=5000, d_model=768, nhead=12, num_encoder_layers=12, d_ffn=3072)
    sd = flat_state_dict(asr)
    sd.update({"LM." + k: v.detach().cpu() for k, v in lm.state_dict().items()})
    n = 2 * 16000
    wav = 0.1 * torch.randn(1, n, generator=torch.Generator().manual_seed(8))
    lens = torch.ones(1)
    enc_ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    T = enc_ref.shape[1]
    scorer = ScorerBuilder(full_scorers=[TransformerLMScorer(language_model=lm, temperature=1.15),
                                         CTCScorer(ctc_fc=asr.mods.ctc_lin, blank_index=0, eos_index=2)],
                           weights={"transformerlm": 0.6, "ctc": 0.4})
    kw = dict(modules=[asr.mods.transformer, asr.mods.seq_lin], bos_index=1, eos_index=2, min_decode_ratio=0.0,
              max_decode_ratio=6.5 / T, beam_size=66, using_eos_threshold=False, length_normalization=True,
              temperature=1.15, scorer=scorer)
    hyps, _, scores, _ = S2STransformerBeamSearcher(**kw)(enc_ref.cuda(), lens.cuda())
    ocfg = O.SearchCfg(beam=66, ctc_weight=0.4, max_decode_ratio=6.5 / T, temperature=1.15, lm_weight=0.6,
                       lm_temperature=1.15)
    hyps_ref, _, scores_ref, _ = O.beam_search(enc_ref, lens, sd, mc, ocfg, lm_cfg=lcfg)
    assert hyps == hyps_ref
    assert float((scores.cpu() - scores_ref).abs().max()) <= 2e-3
    import dataclasses
    k_hyps, _, k_scores, _ = S2STransformerBeamSearcher(**kw, return_topk=True, topk=3)(enc_ref.cuda(), lens.cuda())
    o_hyps, _, o_scores, _ = O.beam_search(enc_ref, lens, sd, mc, dataclasses.replace(ocfg, return_topk=True, topk=3),
                                           lm_cfg=lcfg)
    assert torch.equal(k_hyps.cpu(), o_hyps)
    assert float((k_scores.cpu() - o_scores).abs().max()) <= 2e-3


def test_greedy_beam1_bit_exact_tokens_conformer_l():
    """North-star: bit-exact token ids at greedy / beam = 1 (peaked heads, see above)."""
    from speechbrain_amd.decoders import S2STransformerBeamSearcher, S2STransformerGreedySearcher
    from speechbrain_amd.inference.builders import flat_state_dict

    asr = _asr("L", beam_size=1, ctc_weight=0.0)
    fc, mc = _oracle_cfg("L")
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
    n = 4 * 16000
    wav = 0.1 * torch.randn(2, n, generator=torch.Generator().manual_seed(8))
    lens = torch.tensor([1.0, 0.8])
    wav[1, int(0.8 * n):] = 0
    sd = flat_state_dict(asr)
    enc_ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    T = enc_ref.shape[1]
    ratio = 20.5 / T
    gs = S2STransformerGreedySearcher(modules=[asr.mods.transformer, asr.mods.seq_lin], bos_index=1, eos_index=2,
                                      min_decode_ratio=0.0, max_decode_ratio=ratio)
    hyps, _, _, _ = gs(enc_ref.cuda(), lens.cuda())
    ref, _, _, _ = O.greedy_search(enc_ref, lens, sd, mc, O.SearchCfg(beam=1, max_decode_ratio=ratio))
    assert hyps == ref
    asr.mods.decoder.max_decode_ratio = ratio
    hyps1, _, _, _ = asr.mods.decoder(enc_ref.cuda(), lens.cuda())
    ref1, _, _, _ = O.beam_search(enc_ref, lens, sd, mc, O.SearchCfg(beam=1, max_decode_ratio=ratio))
    assert hyps1 == ref1


def test_properties_at_bench_shape():
    """Size-independent properties at BASELINE's full shape (B = 32, up to 30 s, beam 10 + CTC):
    an utterance decoded inside a batch gives the same tokens as decoded alone with the same
    padding; encoder rows of padded frames are finite; hypotheses have the configured length."""
    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    asr.mods.decoder.check_every = 0
    g = torch.Generator().manual_seed(21)
    B, n = 32, 12 * 16000
    wav = 0.1 * torch.randn(B, n, generator=g)
    lens = torch.linspace(0.5, 1.0, B)
    for i in range(B):
        wav[i, int(lens[i] * n):] = 0
    T = ((1 + n // 160 - 1) // 2 + 1 - 1) // 2 + 1
    asr.mods.decoder.max_decode_ratio = 24.5 / T
    enc = asr.encode_batch(wav, lens)
    assert torch.isfinite(enc).all()
    words, toks = asr.transcribe_batch(wav, lens)
    assert all(len(t) == 23 for t in toks)  # 24 steps, last token stripped (never EOS with random weights)
    for i in (0, 17, 31):
        w1, t1 = asr.transcribe_batch(wav[i:i + 1], lens[i:i + 1])
        assert t1[0] == toks[i]


def test_batches_in_flight_match_sequential_conformer_l():
    """The bench's execution mode at full model size: six batches in flight (host threads, normal- and
    high-priority HIP streams, per-thread helper streams) must give exactly the token ids of sequential
    transcribe_batch calls, and the same ids again on a second concurrent run (no cross-stream races)."""
    from speechbrain_amd.inference.streams import ConcurrentTranscriber

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
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

    ref = []
    for w, l in batches:
        fix_len(asr.mods.decoder, w)
        ref.append(asr.transcribe_batch(w, l)[1])
    ct = ConcurrentTranscriber(asr, streams=6)
    assert ct.transcribe_batches(batches, prepare=fix_len) == ref
    assert ct.transcribe_batches(batches, prepare=fix_len) == ref


def test_conformer_l_unscaled_weights_token_equality_where_margin_allows():
    """UNSCALED random-init Conformer-L (flat posteriors).  The HIP greedy search picks
    a token per step; the oracle, teacher-forced on the same prefix, gives the full-prefix log-probs.  Wherever the
    oracle's own top-1 / top-2 margin exceeds the measured log-prob error, the HIP token MUST be the oracle's
    arg-max (a near-tie may legitimately fall either way); the log-prob of the picked token must agree within 1e-4."""
    from speechbrain_amd.inference.builders import flat_state_dict

    asr = _asr("L", greedy=True)
    fc, mc = _oracle_cfg("L")
    n = 4 * 16000
    wav = 0.1 * torch.randn(2, n, generator=torch.Generator().manual_seed(21))
    lens = torch.tensor([1.0, 0.8])
    wav[1, int(0.8 * n):] = 0
    sd = flat_state_dict(asr)
    enc_ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    T = enc_ref.shape[1]
    steps = 14
    asr.mods.decoder.max_decode_ratio = (steps + 0.5) / T
    hyps, _, scores, _ = asr.mods.decoder(enc_ref.cuda(), lens.cuda())
    enc_lens = torch.round(T * lens).int()
    checked = skipped = 0
    worst = 0.0
    for b, hyp in enumerate(hyps):
        prefix = torch.tensor([[1] + hyp[:-1]]) if len(hyp) > 1 else torch.tensor([[1]])
        dec = O.decode(prefix, enc_ref[b: b + 1], enc_lens[b: b + 1], sd, mc, "Transformer.")
        lp = torch.log_softmax(torch.nn.functional.linear(dec, sd["seq_lin.w.weight"], sd["seq_lin.w.bias"]), -1)[0]
        for t, tok in enumerate(hyp[: lp.shape[0]]):
            top2 = lp[t].topk(2).values
            err = abs(float(scores[b, 0, t]) - float(lp[t, tok]))
            worst = max(worst, err)
            if float(top2[0] - top2[1]) > 5e-4:  # comfortably above the measured error (asserted below)
                assert tok == int(lp[t].argmax()), (b, t, tok, int(lp[t].argmax()))
                checked += 1
            else:
                skipped += 1
    assert worst <= 1e-4, worst
    assert checked >= 10, (checked, skipped)  # the margin rule must not make the test vacuous


def _headline_job(durations, seed):
    """Zero-padded batch of utterances of the given durations (s) + relative lengths, like the bench's batches."""
    g = torch.Generator().manual_seed(seed)
    n = int(max(durations) * 16000)
    wav = torch.zeros(len(durations), n)
    for i, d in enumerate(durations):
        wav[i, : int(d * 16000)] = 0.1 * torch.randn(int(d * 16000), generator=g)
    return wav, torch.tensor([int(d * 16000) / n for d in durations])


def test_headline_shape_beam10_ctc_vs_oracle_end_to_end():
    """the headline's own shape -- Conformer-L, beam 10 + CTC 0.4, utterances of 15-16 s (T' 401,
    the first 40 of their 64 decoding steps; the same test passed at 64 steps in rounds 2-6 and at 24 s / 16 s, T' 601, 96 steps --
    the oracle's Python-loop CTC scorer needs 6 and 14 min for those, so the committed size is the smaller one) -- END TO END (waveform -> token ids: Fbank, CNN, encoder,
    CTC emissions, grouped-search kernels at their long-memory sizes) against the oracle (waveform -> O.encode_batch ->
    O.beam_search).  Peaked output heads (x8) so that fp32 reassociation cannot flip a near-tie:
    token ids exact, scores within 2e-3."""
    from speechbrain_amd.inference.builders import flat_state_dict

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    fc, mc = _oracle_cfg("L")
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
        asr.mods.ctc_lin.w.weight.mul_(8.0)
    wav, lens = _headline_job([16.0, 15.0], seed=41)
    T = ((1 + wav.shape[1] // 160 - 1) // 2 + 1 - 1) // 2 + 1
    # (40 of the 64 steps the bench's rule -- 4 tokens/s -- gives these utterances: the oracle's Python-loop CTC scorer takes 5-9 s per
    #  step at this shape on the box's host cores, and the driver's GPU tier has a 1 200-s limit for the whole suite)
    steps = 40
    assert T >= 400 and steps <= int(round(4.0 * wav.shape[1] / 16000.0))
    ratio = (steps + 0.5) / T
    asr.mods.decoder.max_decode_ratio = ratio
    sd = flat_state_dict(asr)
    _, toks = asr.transcribe_batch(wav, lens)
    enc_ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    hyps_ref, _, scores_ref, _ = O.beam_search(enc_ref, lens, sd, mc, O.SearchCfg(beam=10, ctc_weight=0.4, max_decode_ratio=ratio))
    assert [list(t) for t in toks] == [list(h) for h in hyps_ref]
    assert min(len(t) for t in toks) >= 20  # a degenerate (empty) hypothesis would make the comparison vacuous
    hyps, _, scores, _ = asr.mods.decoder(asr.encode_batch(wav, lens), lens.cuda())
    assert float((scores.cpu() - scores_ref).abs().max()) <= 2e-3


def test_long_utterance_unscaled_weights_margin_rule():
    """VERDICT r2 1(a), second half: UNSCALED random-init Conformer-L at a headline length (20 s, T' 501, 80 steps).
    The HIP greedy search picks a token per step from its own (KV-cached, long-memory) decoder; the oracle, teacher-
    forced on the same prefix, gives the full-prefix log-probs.  Wherever the oracle's own top-1 / top-2 margin exceeds
    5e-4 the HIP token must be the oracle's arg-max; the picked token's log-prob agrees within 2e-4 at every step."""
    from speechbrain_amd.inference.builders import flat_state_dict

    asr = _asr("L", greedy=True)
    fc, mc = _oracle_cfg("L")
    wav, lens = _headline_job([20.0], seed=43)
    sd = flat_state_dict(asr)
    enc_ref = O.encode_batch(wav, lens, sd, fc, mc, torch.zeros(80), torch.ones(80))
    enc = asr.encode_batch(wav, lens)
    assert float((enc.cpu() - enc_ref).abs().max()) <= 2e-4
    T = enc_ref.shape[1]
    steps = 80
    asr.mods.decoder.max_decode_ratio = (steps + 0.5) / T
    hyps, _, scores, _ = asr.mods.decoder(enc, lens.cuda())
    hyp = hyps[0]
    prefix = torch.tensor([[1] + hyp[:-1]])
    dec = O.decode(prefix, enc_ref, torch.round(T * lens).int(), sd, mc, "Transformer.")
    lp = torch.log_softmax(torch.nn.functional.linear(dec, sd["seq_lin.w.weight"], sd["seq_lin.w.bias"]), -1)[0]
    checked, worst = 0, 0.0
    for t, tok in enumerate(hyp[: lp.shape[0]]):
        top2 = lp[t].topk(2).values
        worst = max(worst, abs(float(scores[0, 0, t]) - float(lp[t, tok])))
        if float(top2[0] - top2[1]) > 5e-4:
            assert tok == int(lp[t].argmax()), (t, tok, int(lp[t].argmax()))
            checked += 1
    assert worst <= 2e-4, worst
    assert checked >= 40, checked


def test_headline_mode_grouped_concurrent_equals_sequential_conformer_l():
    """the bench's execution mode at full model size -- ConcurrentTranscriber(streams=8, group=4):
    eight host threads, each encoding four batches and decoding them in ONE grouped search on a high-priority stream --
    on 16 Conformer-L batches of mixed size and length (5-26 s, decode steps = 4 tokens/s x padded seconds, int16 PCM
    from pinned host memory as the bench ships it) gives exactly the token ids of sequential, ungrouped
    transcribe_batch calls (peaked heads: the grouped search runs the decode GEMMs at other row counts, i.e. through
    other kernels with another summation order), and the same ids again on a second run (no cross-stream races)."""
    from speechbrain_amd import native
    from speechbrain_amd.inference.streams import ConcurrentTranscriber

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    asr.mods.decoder.check_every = 0
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
        asr.mods.ctc_lin.w.weight.mul_(8.0)
    g = torch.Generator().manual_seed(57)
    shapes = [(32, 5.0), (32, 7.5), (24, 9.0), (32, 11.0), (16, 13.0), (32, 15.0), (8, 17.0), (32, 19.0),
              (24, 21.0), (12, 23.0), (32, 26.0), (32, 6.0), (20, 8.0), (32, 10.0), (4, 12.0), (32, 14.0)]
    batches = []
    for B, sec in shapes:
        n = int(sec * 16000)
        pcm = (0.1 * torch.randn(B, n, generator=g) * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
        lens = torch.linspace(0.7, 1.0, B)
        for i in range(B):
            pcm[i, int(lens[i] * n):] = 0
        batches.append((pcm.pin_memory(), lens))

    def fix_len(searcher, wavs):
        T = ((1 + wavs.shape[1] // 160 - 1) // 2 + 1 - 1) // 2 + 1
        searcher.max_decode_ratio = (int(round(4.0 * wavs.shape[1] / 16000.0)) + 0.5) / T

    ref = []
    for w, l in batches:
        fix_len(asr.mods.decoder, w)
        ref.append(asr.transcribe_batch(native.pcm16_to_f32(w.cuda()), l.cuda())[1])
    assert all(len(t) > 0 for r in ref for t in r)
    ct = ConcurrentTranscriber(asr, streams=8, group=4)
    got = ct.transcribe_batches(batches, prepare=fix_len)
    assert got == ref
    assert ct.transcribe_batches(batches, prepare=fix_len) == ref


def test_whisper_large_v3_shape_encoder_vs_reference_and_hf():
    """BASELINE.json configs[4] at the large-v3 SHAPE (d 1280, 20 heads of 64, 128 mel bins, 1500 positions, ffn 5120;
    2 encoder layers): log-mel + encoder on 2 x 30 s against (a) strided samples of the REFERENCE wrapper's outputs
    (tests/golden/whisper_large_shape.npz, oracle/make_golden.py --whisper-large-only; the 157 MB of weights are
    transformers' own seeded initialisation, rebuilt here) and (b) the transformers model itself, run on the host at
    test time.  fp32: mel 2e-4, encoder 5e-4 absolute on outputs up to 4.5.  Then the opt-in
    reduced-precision operand paths (bf16 incl. attention, fp16, fp8 e4m3) at the same shape, each within its stated
    tolerance of the fp32 output."""
    import os

    import numpy as np

    tf = pytest.importorskip("transformers")
    from speechbrain_amd import native
    from speechbrain_amd.integrations.huggingface.whisper import Whisper

    import emu_utils

    emu_utils.detach()
    native.load()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "whisper_large_shape.npz"))
    cfg = dict(vocab_size=51866, num_mel_bins=128, d_model=1280, encoder_layers=2, encoder_attention_heads=20,
               encoder_ffn_dim=5120, decoder_layers=1, decoder_attention_heads=20, decoder_ffn_dim=5120,
               max_source_positions=1500, max_target_positions=448)
    torch.manual_seed(21)
    hf = tf.WhisperModel(tf.WhisperConfig(**cfg)).eval()
    same_init = abs(float(hf.encoder.layers[0].fc1.weight.double().sum()) - float(g["first_param_sum"])) < 1e-6
    w = Whisper.from_config(cfg, encoder_only=True)
    w.model.load_state_dict({k: v.float() for k, v in hf.state_dict().items() if k.startswith("encoder.")}, strict=True)
    w = w.to("cuda:0").eval()
    gen = torch.Generator().manual_seed(22)
    wav = torch.stack([0.1 * torch.randn(480000, generator=gen),
                       torch.cat([0.05 * torch.randn(300000, generator=gen), torch.zeros(180000)])])
    mel = w._get_mel(wav.cuda())
    assert mel.shape == (2, 128, 3000)
    enc = w.forward_encoder(mel)
    assert enc.shape == (2, 1500, 1280)
    with torch.no_grad():
        enc_hf = hf.encoder(mel.cpu()).last_hidden_state
    assert float((enc.cpu() - enc_hf).abs().max()) <= 5e-4
    if same_init:  # the reference wrapper's own mel and encoder output (same transformers build => same weights)
        assert float((mel.cpu()[:, ::8, ::50] - torch.from_numpy(g["mel_sample"])).abs().max()) <= 2e-4
        assert float((enc.cpu()[:, ::25, ::32] - torch.from_numpy(g["enc_sample"])).abs().max()) <= 5e-4
    # the opt-in reduced-precision GEMM operands at this shape, against the fp32 output (absolute max, relative RMS):
    # bf16 (GEMMs + attention) 0.15 / 1 %; fp16 (GEMMs) 0.05 / 0.3 %; fp8 e4m3 with per-tensor scales (GEMMs) 1.0 / 8 %
    got = {}
    for prec, (tol_abs, tol_rms) in (("bf16", (0.15, 1e-2)), ("fp16", (0.05, 3e-3)), ("fp8", (1.0, 8e-2))):
        with native.precision_scope(prec):
            e = w.forward_encoder(mel)
        err = (e - enc).float()
        got[prec] = (float(err.abs().max()), float(err.pow(2).mean().sqrt() / enc.pow(2).mean().sqrt()))
        print(f"whisper large-v3-shape encoder, {prec} GEMM operands vs fp32: max |d| {got[prec][0]:.4f}, relative RMS {got[prec][1]:.5f}")
        assert got[prec][0] <= tol_abs and got[prec][1] <= tol_rms, (prec, got[prec])


def test_whisper_large_v3_full_depth_reduced_precision_through_the_interface():
    """BASELINE configs[4] at FULL depth (the fp8 tolerance was asserted at 2 layers only, the 32-layer figure came
    from an untested bench print): a large-v3 shaped encoder with all 32 layers (random weights), 2 x 30 s, driven through
    WhisperASR(run_opts={"precision": p}).encode_batch.  Relative RMS of the encoder output against the fp32 interface:
    bf16 <= 1 %, fp8 (e4m3 activation pipeline on v_mfma_scale_f32_32x32x64_f8f6f4) <= 8 % -- the same bounds as at 2 layers
    (the residual stream stays fp32, so the error does not compound with depth: measured 0.33 % / 5.2 % in the round-4 bench)."""
    from speechbrain_amd import native
    from speechbrain_amd.inference.ASR import WhisperASR
    from speechbrain_amd.integrations.huggingface.whisper import Whisper

    cfg = dict(num_mel_bins=128, d_model=1280, encoder_layers=32, encoder_attention_heads=20, encoder_ffn_dim=5120,
               max_source_positions=1500, decoder_layers=0, decoder_attention_heads=20, decoder_ffn_dim=5120,
               vocab_size=51866, max_target_positions=448)
    w = Whisper.from_config(cfg, encoder_only=True, seed=6).cuda().eval()
    wav = (0.1 * torch.randn(2, 480000, generator=torch.Generator().manual_seed(3))).cuda()
    hp = {"language": "en", "sample_rate": 16000, "whisper": w}
    calls = {"fp8a": 0}
    g0 = native.gemm_nt_fp8a

    def counted(*a, **k):
        calls["fp8a"] += 1
        return g0(*a, **k)

    out = {}
    native.gemm_nt_fp8a = counted
    try:
        with torch.no_grad():
            for prec in ("fp32", "bf16", "fp8"):
                asr = WhisperASR(modules={"whisper": w, "decoder": torch.nn.Identity()}, hparams=hp, run_opts={"device": "cuda:0", "precision": prec})
                out[prec] = asr.encode_batch(wav, torch.ones(2))
    finally:
        native.gemm_nt_fp8a = g0
    assert calls["fp8a"] == 4 * 32  # the four contractions of every layer on the fp8 instruction, selected by run_opts alone
    rms = float(out["fp32"].pow(2).mean().sqrt())
    rel = {p: float((out[p] - out["fp32"]).pow(2).mean().sqrt()) / rms for p in ("bf16", "fp8")}
    print(f"whisper large-v3 shape, 32 layers, relative RMS vs fp32: {rel}")
    assert torch.isfinite(out["fp8"]).all() and 0.0 < rel["bf16"] <= 1e-2 and 0.0 < rel["fp8"] <= 8e-2, rel


@pytest.mark.timeout(300)
def test_persistent_step_from_several_threads_at_once_conformer_l():
    """The persistent few-row decoding step is a COOPERATIVE launch whose workgroups wait for each other at grid barriers:
    four host threads, each on its own stream, run single-utterance searches (beam 10 + CTC, Conformer-L) at the same time while
    a fifth keeps the chip busy with large contractions -- every result must equal the sequential one and nothing may hang
    (tools/coop_concurrency_check.py is the same check as a script; profiles/r05_j_*: 4 and 8 threads, 0 differing results)."""
    import threading

    from speechbrain_amd import native

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    with torch.no_grad():
        asr.mods.seq_lin.w.weight.mul_(8.0)
        asr.mods.ctc_lin.w.weight.mul_(8.0)
    dec = asr.mods.decoder
    dec.max_decode_ratio = 0.1
    g = torch.Generator().manual_seed(1)
    encs = [(torch.randn(1, 120 + 17 * k, 512, generator=g).cuda(), torch.ones(1).cuda()) for k in range(4)]
    native.prof_reset()
    native.prof_enable(True)
    with torch.no_grad():
        ref = [dec(e, l)[0] for e, l in encs]
    native.prof_enable(False)
    assert "decoder_step_persist" in native.prof_report()
    torch.cuda.synchronize()
    stop, bad = threading.Event(), []

    def load():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        a, w = torch.randn(8192, 512, device="cuda"), torch.randn(2048, 512, device="cuda")
        with torch.cuda.stream(s), torch.no_grad():
            while not stop.is_set():
                for _ in range(4):
                    native.gemm_nt(a, w)
                s.synchronize()

    def worker(k):
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s), torch.no_grad():
            for r in range(6):
                if dec(*encs[k])[0] != ref[k]:
                    bad.append((k, r))
            s.synchronize()

    lt = threading.Thread(target=load)
    lt.start()
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    stop.set()
    lt.join()
    torch.cuda.synchronize()
    assert not bad, bad


def test_in_kernel_handoffs_under_uneven_load_conformer_l():
    """Every in-launch hand-off of the path (stream-K partial tiles -> last-arriver fix-up in the persistent / split-operand
    contractions, split-K tickets of the decode GEMMs, the cross-attention runs' last-arriver merge) must give
    bit-identical results however the chip is loaded: MI355X_MICROARCH.md asks for hand-offs to be tested "under UNEVEN
    load, consumer L1-warm, checking every word" -- an idle chip hides a missing release / acquire.  A second host thread
    keeps a stream busy with bursts of bandwidth- and matrix-heavy launches of changing size while (a) the encoder of one
    batch (24 x 7.5 s: GPUTEST_r03's differing batch) and (b) its beam search run again and again on another stream; every
    word of the encoder output, every token id and every score must equal the first, unloaded, run."""
    import threading

    from speechbrain_amd import native

    asr = _asr("L", beam_size=10, ctc_weight=0.4)
    asr.mods.decoder.check_every = 0
    n = int(7.5 * 16000)
    wav = 0.1 * torch.randn(24, n, generator=torch.Generator().manual_seed(33))
    lens = torch.linspace(0.6, 1.0, 24)
    for i in range(24):
        wav[i, int(lens[i] * n):] = 0
    wav, lens = wav.cuda(), lens.cuda()
    T = ((1 + n // 160 - 1) // 2 + 1 - 1) // 2 + 1
    asr.mods.decoder.max_decode_ratio = 20.5 / T
    enc0 = asr.encode_batch(wav, lens).clone()
    tok0, ln0, sc0, lp0, _ = asr.mods.decoder.search_device(enc0, lens)
    tok0, sc0, lp0 = tok0.clone(), sc0.clone(), lp0.clone()
    torch.cuda.synchronize()

    stop = threading.Event()

    def load():  # bursts of different weight with pauses between them: the chip's load changes under the measured stream
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        g = torch.Generator(device="cuda").manual_seed(1)
        with torch.cuda.stream(s):
            big = torch.randn(64 << 20, device="cuda", generator=g)
            a = torch.randn(6000, 512, device="cuda", generator=g)
            w = torch.randn(2048, 512, device="cuda", generator=g)
            k = 0
            while not stop.is_set():
                k += 1
                if k % 3 == 0:
                    big[: (8 << 20) * (1 + k % 8)].mul_(1.0001)   # 32 .. 256 MB of HBM traffic
                elif k % 3 == 1:
                    for _ in range(1 + k % 5):
                        native.gemm_nt(a[: 2048 + 512 * (k % 8)], w)  # matrix bursts (split-operand / persistent kernels)
                else:
                    s.synchronize()                                 # a pause: the other stream has the chip to itself
            s.synchronize()

    th = threading.Thread(target=load)
    th.start()
    try:
        work = torch.cuda.Stream(priority=-1)
        with torch.cuda.stream(work):
            for rep in range(25):
                enc = asr.encode_batch(wav, lens)
                assert torch.equal(enc, enc0), f"encoder output differs under load (rep {rep}): {int((enc != enc0).sum())} words"
                tok, ln, sc, lp, _ = asr.mods.decoder.search_device(enc0, lens)
                assert torch.equal(tok, tok0) and torch.equal(sc, sc0) and torch.equal(lp, lp0), f"search differs under load (rep {rep})"
    finally:
        stop.set()
        th.join()
