"""Programmatic equivalent of the recipe YAML (recipes/LibriSpeech/ASR/transformer/hparams/
conformer_{large,small}.yaml:66-70,110-118,138-251) wired the way an inference YAML wires
EncoderDecoderASR (templates/speech_recognition/ASR/inference.yaml:123-135)."""
import torch

from speechbrain_amd.decoders import CTCScorer, S2STransformerBeamSearcher, S2STransformerGreedySearcher, ScorerBuilder
from speechbrain_amd.inference.ASR import EncoderDecoderASR
from speechbrain_amd.lobes.features import Fbank
from speechbrain_amd.lobes.models.convolution import ConvolutionFrontEnd
from speechbrain_amd.lobes.models.transformer.TransformerASR import TransformerASR
from speechbrain_amd.nnet.containers import LengthsCapableSequential
from speechbrain_amd.nnet.linear import Linear
from speechbrain_amd.processing.features import InputNormalization

SIZES = {
    # d_model, nhead, d_ffn, enc, dec, n_fft, win_length (ms)
    "L": dict(d_model=512, nhead=8, d_ffn=2048, n_enc=12, n_dec=6, n_fft=512, win_length=32),
    "S": dict(d_model=144, nhead=4, d_ffn=1024, n_enc=12, n_dec=4, n_fft=400, win_length=25),
}


class IdTokenizer:
    """Stand-in for the SentencePiece model when no tokenizer.ckpt is available (no network here)."""

    def decode_ids(self, ids):
        return " ".join(str(int(i)) for i in ids)


def build_modules(size="L", vocab=5000, seed=0, **override):
    """Random-initialised (torch.manual_seed(seed)) modules with the reference's initialisers."""
    cfg = {**SIZES[size], **override} if isinstance(size, str) else {**size, **override}
    torch.manual_seed(seed)
    cnn = ConvolutionFrontEnd(input_shape=(8, 10, 80), num_blocks=2, num_layers_per_block=1, out_channels=(64, 32),
                              kernel_sizes=(3, 3), strides=(2, 2), residuals=(False, False))
    transformer = TransformerASR(input_size=640, tgt_vocab=vocab, d_model=cfg["d_model"], nhead=cfg["nhead"],
                                 num_encoder_layers=cfg["n_enc"], num_decoder_layers=cfg["n_dec"], d_ffn=cfg["d_ffn"],
                                 dropout=0.1, activation=torch.nn.GELU, encoder_module="conformer",
                                 attention_type=cfg.get("attention_type", "RelPosMHAXL"), normalize_before=True,
                                 causal=False)
    ctc_lin = Linear(input_size=cfg["d_model"], n_neurons=vocab)
    seq_lin = Linear(input_size=cfg["d_model"], n_neurons=vocab)
    normalize = InputNormalization(norm_type="global", update_until_epoch=4)
    normalize.glob_mean, normalize.glob_std, normalize.count = torch.zeros(80), torch.ones(80), 1
    fbank = Fbank(sample_rate=16000, n_fft=cfg["n_fft"], n_mels=80, win_length=cfg["win_length"])
    return dict(CNN=cnn, Transformer=transformer, ctc_lin=ctc_lin, seq_lin=seq_lin, normalize=normalize,
                compute_features=fbank, cfg=cfg)


def build_asr(size="L", vocab=5000, seed=0, beam_size=10, ctc_weight=0.4, max_decode_ratio=1.0, min_decode_ratio=0.0,
              using_eos_threshold=False, greedy=False, device=None, tokenizer=None, modules=None, **override):
    """EncoderDecoderASR with the recipe's ``valid_search`` (beam 10 + CTC 0.4, conformer_large.yaml:225-239)."""
    m = modules or build_modules(size, vocab, seed, **override)
    encoder = LengthsCapableSequential(compute_features=m["compute_features"], normalize=m["normalize"], model=m["CNN"])
    if greedy:
        decoder = S2STransformerGreedySearcher(modules=[m["Transformer"], m["seq_lin"]], bos_index=1, eos_index=2,
                                               min_decode_ratio=min_decode_ratio, max_decode_ratio=max_decode_ratio)
    else:
        scorer = None
        if ctc_weight > 0:
            scorer = ScorerBuilder(full_scorers=[CTCScorer(ctc_fc=m["ctc_lin"], blank_index=0, eos_index=2)],
                                   weights={"ctc": ctc_weight})
        decoder = S2STransformerBeamSearcher(modules=[m["Transformer"], m["seq_lin"]], bos_index=1, eos_index=2,
                                             min_decode_ratio=min_decode_ratio, max_decode_ratio=max_decode_ratio,
                                             beam_size=beam_size, using_eos_threshold=using_eos_threshold,
                                             length_normalization=True, scorer=scorer)
    run_opts = {"device": device} if device is not None else None
    asr = EncoderDecoderASR(
        modules={"encoder": encoder, "transformer": m["Transformer"], "decoder": decoder, "ctc_lin": m["ctc_lin"],
                 "seq_lin": m["seq_lin"]},
        hparams={"tokenizer": tokenizer or IdTokenizer(), "transformer_beam_search": True}, run_opts=run_opts)
    return asr


def flat_state_dict(asr):
    """One flat state_dict of the ASR modules under the prefixes CNN. / Transformer. / seq_lin. / ctc_lin. (CPU tensors)."""
    sd = {}
    for pfx, mod in (("CNN.", asr.mods.encoder["model"]), ("Transformer.", asr.mods.transformer),
                     ("seq_lin.", asr.mods.seq_lin), ("ctc_lin.", asr.mods.ctc_lin)):
        for k, v in mod.state_dict().items():
            sd[pfx + k] = v.detach().cpu()
    return sd
