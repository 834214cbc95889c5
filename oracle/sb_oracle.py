"""CPU oracle for the EncoderDecoderASR hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional (state_dict in, tensors out) restatement, in plain
fp32 PyTorch-CPU ops, of the algorithm SpeechBrain's own modules run for the
path  STFT/Fbank -> InputNormalization -> ConvolutionFrontEnd -> TransformerASR
.encode (Conformer, RelPosMHAXL or RoPEMHA) -> S2STransformer{Beam,Greedy}Searcher
(+ CTCScorer / CTCPrefixScore, TransformerLMScorer / TransformerLM, return_topk).

* It is the checker, never the product: only ``tests/``,
  ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
  it.  Nothing under ``speechbrain_amd/`` imports this package.
* Parity pinning: ``oracle/make_golden.py`` runs the *real* reference
  (``/root/reference`` on sys.path + 3 stub modules) and this file on the same
  seeded weights/inputs, asserts agreement, and writes ``tests/golden/*.npz``.
  ``tests/test_oracle_golden.py`` re-checks this file against those vectors on
  machines without the reference.  The reference's own known-answer tests for
  the path (tests/unittests/test_features.py:57-118, doctests cited below) are
  replayed in ``tests/test_oracle_known_answers.py``.
* Each function cites the reference file:line (relative to
  ``/root/reference/speechbrain``) whose behaviour it follows.  The structure
  (pure functions over a flat dict of weights) is ours; the arithmetic order is
  the reference's wherever it matters for fp32 parity (e.g. no KV cache, the
  CTC prefix scorer's frame loop and finite -1e20 sentinel).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclass
class FbankCfg:
    """lobes/features.py:98-145 (Fbank ctor) defaults the recipe overrides."""

    sample_rate: int = 16000
    n_fft: int = 400
    n_mels: int = 40
    win_length_ms: float = 25
    hop_length_ms: float = 10
    f_min: float = 0.0
    f_max: Optional[float] = None
    top_db: float = 80.0
    amin: float = 1e-10

    @property
    def win(self) -> int:  # processing/features.py:131-133
        return int(round((self.sample_rate / 1000.0) * self.win_length_ms))

    @property
    def hop(self) -> int:  # processing/features.py:134-136
        return int(round((self.sample_rate / 1000.0) * self.hop_length_ms))


@dataclass
class ModelCfg:
    """TransformerASR.py:247-273 arguments on the path."""

    d_model: int = 512
    nhead: int = 8
    num_encoder_layers: int = 12
    num_decoder_layers: int = 6
    d_ffn: int = 2048
    vocab: int = 5000
    kernel_size: int = 31
    input_size: int = 640
    attention_type: str = "RelPosMHAXL"  # or "RoPEMHA" (conformer_large.yaml:158 of the current recipe)


@dataclass
class SearchCfg:
    """decoders/seq2seq.py:752-768 + scorer weights (scorer.py:1163-1200)."""

    bos: int = 1
    eos: int = 2
    blank: int = 0
    beam: int = 10
    min_decode_ratio: float = 0.0
    max_decode_ratio: float = 1.0
    ctc_weight: float = 0.0
    temperature: float = 1.0
    length_normalization: bool = True
    using_eos_threshold: bool = False
    eos_threshold: float = 1.5
    minus_inf: float = -1e20
    topk: int = 1  # with return_topk (seq2seq.py:757-760): the searcher returns padded [B,topk,...] tensors
    return_topk: bool = False
    lm_weight: float = 0.0  # ScorerBuilder weights["transformerlm"] (scorer.py:1163-1200)
    lm_temperature: float = 1.0  # TransformerLMScorer.temperature (scorer.py:504-508)


@dataclass
class LMCfg:
    """TransformerLM.py:68-85 arguments on the path (encoder-only, regularMHA)."""

    vocab: int = 5000
    d_model: int = 768
    nhead: int = 12
    num_encoder_layers: int = 12
    d_ffn: int = 3072
    normalize_before: bool = False
    activation: str = "gelu"
    pad_idx: int = 0


# --------------------------------------------------------------------------
# a2-a6: STFT / Fbank / InputNormalization
# --------------------------------------------------------------------------
def hamming_window(win: int) -> Tensor:
    """torch.hamming_window default (periodic), processing/features.py:115,139."""
    n = torch.arange(win, dtype=torch.float32)
    return 0.54 - 0.46 * torch.cos(2.0 * math.pi * n / win)


def stft_power(wav: Tensor, cfg: FbankCfg) -> Tensor:
    """[B,N] -> [B,T,n_fft/2+1] power spectrum (re^2+im^2).

    processing/features.py:141-188 (torch.stft center=True, pad_mode constant,
    onesided, not normalised) followed by spectral_magnitude(power=1)
    (:341-378).  Window shorter than n_fft is centred (torch.stft semantics).
    """
    n_fft, win, hop = cfg.n_fft, cfg.win, cfg.hop
    w = hamming_window(win)
    if win < n_fft:
        left = (n_fft - win) // 2
        w = F.pad(w, (left, n_fft - win - left))
    x = F.pad(wav.float(), (n_fft // 2, n_fft // 2))
    frames = x.unfold(1, n_fft, hop)  # [B,T,n_fft]
    spec = torch.fft.rfft(frames * w, dim=-1)
    return spec.real.square() + spec.imag.square()


def mel_filterbank(cfg: FbankCfg) -> Tensor:
    """[n_fft/2+1, n_mels] triangular filters.

    processing/features.py:487-510 (mel points, band, central freq) and
    :620-647 (_triangular_filters: both slopes use the same band).
    """
    f_max = cfg.f_max if cfg.f_max is not None else cfg.sample_rate / 2
    n_stft = cfg.n_fft // 2 + 1

    def to_mel(hz):
        return 2595 * math.log10(1 + hz / 700)

    mel = torch.linspace(to_mel(cfg.f_min), to_mel(f_max), cfg.n_mels + 2)
    hz = 700 * (10 ** (mel / 2595) - 1)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    all_freqs = torch.linspace(0, cfg.sample_rate // 2, n_stft)
    slope = (all_freqs[None, :] - f_central[:, None]) / band[:, None]
    fb = torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0)
    return fb.t().contiguous()


def fbank(wav: Tensor, cfg: FbankCfg) -> Tensor:
    """lobes/features.py:147-169 with deltas/context off: [B,N] -> [B,T,n_mels].

    dB conversion and the per-utterance (max over all frames x mels, padded
    frames included) - top_db floor: processing/features.py:736-759.
    """
    p = stft_power(wav, cfg)
    fb = torch.matmul(p, mel_filterbank(cfg))
    x_db = 10.0 * torch.log10(torch.clamp(fb, min=cfg.amin))
    floor = x_db.amax(dim=(-2, -1)) - cfg.top_db
    return torch.maximum(x_db, floor.view(-1, 1, 1))


def input_norm_global(x: Tensor, mean: Tensor, std: Tensor, eps: float = 1e-10) -> Tensor:
    """InputNormalization eval, norm_type="global" (features.py:1404-1455)."""
    return (x - mean.view(1, 1, -1)) / std.view(1, 1, -1).clamp(min=eps)


def padding_mask(T: int, lengths: Tensor, eps: float = 1e-6) -> Tensor:
    """make_padding_mask (features.py:1554-1615): True = valid, [B,T]."""
    return torch.arange(T)[None, :] < (lengths * T - eps)[:, None]


def input_norm_sentence(x: Tensor, lengths: Tensor, eps: float = 1e-10) -> Tensor:
    """InputNormalization norm_type="sentence" (features.py:1436-1437,1478-1486)."""
    m = padding_mask(x.shape[1], lengths).unsqueeze(-1)
    n = m.sum(1, keepdim=True)
    mean = (x * m).sum(1, keepdim=True) / n
    var = ((x - mean) * m).square().sum(1, keepdim=True) / n
    return (x - mean) / var.sqrt().clamp(min=eps)


# --------------------------------------------------------------------------
# a7: ConvolutionFrontEnd
# --------------------------------------------------------------------------
def conv_frontend(x: Tensor, sd: SD, prefix: str = "", num_blocks: int = 2) -> Tensor:
    """[B,T,F] -> [B,T',F',C].

    lobes/models/convolution.py:162-198,311-317; nnet/CNN.py:654-751 (transpose
    to [B,C,F,T], reflect pad floor(k/2) because stride>1, :1510-1536);
    LayerNorm over (F',C) eps 1e-5 (nnet/normalization.py:185-242);
    LeakyReLU(0.01).
    """
    h = x.unsqueeze(-1)  # [B,T,F,C=1]
    for i in range(num_blocks):
        w = sd[f"{prefix}convblock_{i}.convs.conv_0.conv.weight"]
        b = sd[f"{prefix}convblock_{i}.convs.conv_0.conv.bias"]
        g = sd[f"{prefix}convblock_{i}.convs.norm_0.norm.weight"]
        be = sd[f"{prefix}convblock_{i}.convs.norm_0.norm.bias"]
        z = h.permute(0, 3, 2, 1)  # [B,C,F,T]
        z = F.pad(z, (1, 1, 1, 1), mode="reflect")
        z = F.conv2d(z, w, b, stride=2)
        h = z.permute(0, 3, 2, 1)  # [B,T',F',C]
        h = F.layer_norm(h, h.shape[2:], g, be, 1e-5)
        h = F.leaky_relu(h, 0.01)
    return h


# --------------------------------------------------------------------------
# a8-a13: TransformerASR.encode with the Conformer encoder
# --------------------------------------------------------------------------
def relpos_table(T: int, d: int) -> Tensor:
    """RelPosEncXL.make_pe (nnet/attention.py:347-427): [2T-1, d].

    Row T-1+r and row T-1-r are identical ([sin(r f), cos(r f)] interleaved).
    """
    inv = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(-1)
    pe = torch.empty(T, d)
    pe[:, 0::2] = torch.sin(pos * inv)
    pe[:, 1::2] = torch.cos(pos * inv)
    return torch.cat([torch.flip(pe, (0,)), pe[1:]], dim=0)


def rel_shift(x: Tensor) -> Tensor:
    """RelPosMHAXL.rel_shift (attention.py:537-553): [B,H,T,2T-1] -> [B,H,T,T]."""
    b, h, q, p = x.shape
    x = F.pad(x, (1, 0)).view(b, h, -1, q)[:, :, 1:].view(b, h, q, p)
    return x[..., : p // 2 + 1]


def relpos_mha(x: Tensor, pos: Tensor, sd: SD, pfx: str, H: int, key_pad: Optional[Tensor]) -> Tensor:
    """Self-attention branch of RelPosMHAXL.forward (attention.py:555-742).

    x [B,T,E]; pos [2T-1,E]; key_pad [B,T] True = padded.
    """
    B, T, E = x.shape
    Dh = E // H
    qkv = F.linear(x, sd[pfx + "in_proj_weight"]).view(B, T, H, 3 * Dh)
    q, k, v = qkv.chunk(3, dim=-1)  # per-head interleave (:623-626)
    u = sd[pfx + "pos_bias_u"].view(1, 1, H, Dh)  # (Dh,H) *viewed* as (H,Dh)
    vb = sd[pfx + "pos_bias_v"].view(1, 1, H, Dh)
    p = F.linear(pos, sd[pfx + "linear_pos.weight"]).view(1, -1, H, Dh)
    s = 1.0 / math.sqrt(E)  # scale uses embed_dim (:521)
    ac = torch.matmul(((q + u) * s).transpose(1, 2), k.permute(0, 2, 3, 1))
    bd = rel_shift(torch.matmul(((q + vb) * s).transpose(1, 2), p.permute(0, 2, 3, 1)))
    score = ac + bd
    if key_pad is not None:
        score = score.masked_fill(key_pad.view(B, 1, 1, T), float("-inf"))
    att = F.softmax(score, dim=-1, dtype=torch.float32)
    if key_pad is not None:
        att = att.masked_fill(key_pad.view(B, 1, 1, T), 0.0)
    o = torch.matmul(att, v.transpose(1, 2)).transpose(1, 2).reshape(B, T, E)
    return F.linear(o, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def rope_tables(T: int, Dh: int):
    """PrecomputedRoPESinusoids (nnet/attention.py:955-1053): cosines, signed sines [T,Dh]."""
    angles = torch.exp(torch.arange(0, Dh, 2, dtype=torch.float32) * -(math.log(10000.0) / Dh))
    ta = torch.outer(torch.arange(0, T, dtype=torch.float32), angles)
    cos = torch.stack([torch.cos(ta)] * 2, dim=-1).reshape(T, Dh)
    sin = torch.stack([torch.sin(ta)] * 2, dim=-1).reshape(T, Dh)
    sin = ((-1) ** torch.arange(Dh, dtype=torch.float32)) * -sin
    return cos, sin


def rope_rotate(x: Tensor) -> Tensor:
    """_rope_rotate (attention.py:1167-1188): x [B,T,H,Dh], pairs (2i, 2i+1) rotated by t * theta_i."""
    _, T, _, Dh = x.shape
    cos, sin = rope_tables(T, Dh)
    swap = torch.arange(Dh).view(-1, 2).flip(-1).reshape(-1)
    return x * cos.unsqueeze(1) + x[..., swap] * sin.unsqueeze(1)


def rope_mha(x: Tensor, sd: SD, pfx: str, H: int, key_pad: Optional[Tensor]) -> Tensor:
    """RoPEMHA.forward, self-attention branch (attention.py:1284-1392): bias-free stacked in_proj viewed
    per head as (q|k|v), rotary q/k, scale 1/sqrt(embed_dim) (:1272), key padding mask, out_proj."""
    B, T, E = x.shape
    Dh = E // H
    q, k, v = F.linear(x, sd[pfx + "in_proj_weight"]).view(B, T, H, 3 * Dh).chunk(3, dim=-1)
    q, k = rope_rotate(q), rope_rotate(k)
    sc = torch.einsum("bihd,bjhd->bhij", q, k) * (1.0 / math.sqrt(E))
    if key_pad is not None:
        sc = sc.masked_fill(key_pad.view(B, 1, 1, T), float("-inf"))
    o = torch.einsum("bhij,bjhd->bihd", F.softmax(sc, dim=-1), v).reshape(B, T, E)
    return F.linear(o, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def _ffn(x: Tensor, sd: SD, pfx: str, act) -> Tensor:
    """PositionalwiseFeedForward (attention.py:915-947)."""
    h = act(F.linear(x, sd[pfx + "ffn.0.weight"], sd[pfx + "ffn.0.bias"]))
    return F.linear(h, sd[pfx + "ffn.3.weight"], sd[pfx + "ffn.3.bias"])


def _ln(x: Tensor, sd: SD, pfx: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + "weight"], sd[pfx + "bias"], eps)


def conv_module(x: Tensor, sd: SD, pfx: str, pad_mask: Optional[Tensor], ksize: int) -> Tensor:
    """ConvolutionModule non-causal branch (Conformer.py:315-330; ctor :105-157)."""
    d = x.shape[-1]
    h = _ln(x, sd, pfx + "layer_norm.", 1e-5).transpose(1, 2)
    h = F.conv1d(h, sd[pfx + "bottleneck.0.weight"], sd[pfx + "bottleneck.0.bias"])
    h = F.glu(h, dim=1)
    h = F.conv1d(h, sd[pfx + "conv.weight"], sd[pfx + "conv.bias"], padding=(ksize - 1) // 2, groups=d)
    h = h.transpose(1, 2)
    h = F.silu(_ln(h, sd, pfx + "after_conv.0.", 1e-5))
    h = F.linear(h, sd[pfx + "after_conv.2.weight"], sd[pfx + "after_conv.2.bias"])
    if pad_mask is not None:
        h = h.masked_fill(pad_mask.unsqueeze(-1), 0.0)
    return h


def conformer_layer(x, pos, sd, pfx, H, key_pad, ksize):
    """ConformerEncoderLayer.forward (Conformer.py:451-499); pos = None selects RoPEMHA (:414-419)."""
    x = x + 0.5 * _ffn(_ln(x, sd, pfx + "ffn_module1.0.", 1e-5), sd, pfx + "ffn_module1.1.", F.silu)
    skip = x
    h = _ln(x, sd, pfx + "norm1.norm.", 1e-5)
    if pos is None:
        x = rope_mha(h, sd, pfx + "mha_layer.", H, key_pad) + skip
    else:
        x = relpos_mha(h, pos, sd, pfx + "mha_layer.", H, key_pad) + skip
    x = x + conv_module(x, sd, pfx + "convolution_module.", key_pad, ksize)
    y = x + 0.5 * _ffn(_ln(x, sd, pfx + "ffn_module2.0.", 1e-5), sd, pfx + "ffn_module2.1.", F.silu)
    return _ln(y, sd, pfx + "norm2.norm.", 1e-5)


def length_to_mask(length: Tensor, max_len: Optional[int] = None) -> Tensor:
    """dataio/dataio.py:803-848: [B] -> [B,max_len] bool, True = valid."""
    if max_len is None:
        max_len = int(length.max().item())
    return torch.arange(max_len)[None, :] < length[:, None]


def encode(src: Tensor, wav_lens: Optional[Tensor], sd: SD, cfg: ModelCfg, pfx: str = "", return_layers: bool = False):
    """TransformerASR.encode (TransformerASR.py:475-544): [B,T',F',C] -> [B,T',d].

    Key padding mask from round(len*T') (:146-149); custom_src_module Linear;
    ConformerEncoder (Conformer.py:705-778) with final LayerNorm eps 1e-6.
    """
    if src.dim() == 4:
        src = src.reshape(src.shape[0], src.shape[1], -1)
    B, T, _ = src.shape
    key_pad = None
    if wav_lens is not None:
        key_pad = ~length_to_mask(torch.round(wav_lens * T), T)
    x = F.linear(src, sd[pfx + "custom_src_module.layers.0.w.weight"], sd[pfx + "custom_src_module.layers.0.w.bias"])
    pos = None if cfg.attention_type == "RoPEMHA" else relpos_table(T, cfg.d_model)  # TransformerASR.py:519-528
    layers = []
    for l in range(cfg.num_encoder_layers):
        x = conformer_layer(x, pos, sd, f"{pfx}encoder.layers.{l}.", cfg.nhead, key_pad, cfg.kernel_size)
        if return_layers:
            layers.append(x)
    out = _ln(x, sd, pfx + "encoder.norm.norm.", 1e-6)
    return (out, layers) if return_layers else out


# --------------------------------------------------------------------------
# a16-a17: TransformerASR.decode (full prefix, no KV cache)
# --------------------------------------------------------------------------
def abs_pos_encoding(L: int, d: int) -> Tensor:
    """PositionalEncoding (Transformer.py:252-303): rows 0..L-1."""
    pos = torch.arange(0, L).unsqueeze(1).float()
    den = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe = torch.zeros(L, d)
    pe[:, 0::2] = torch.sin(pos * den)
    pe[:, 1::2] = torch.cos(pos * den)
    return pe


def _mha(q_in, kv_in, sd, pfx, H, attn_mask=None, key_pad=None):
    """torch.nn.MultiheadAttention as wrapped at nnet/attention.py:778-886
    (stacked in_proj + bias, scale 1/sqrt(head_dim), float -inf causal mask,
    bool key padding mask)."""
    B, L, E = q_in.shape
    S = kv_in.shape[1]
    Dh = E // H
    W, bias = sd[pfx + "att.in_proj_weight"], sd[pfx + "att.in_proj_bias"]
    q = F.linear(q_in, W[:E], bias[:E]).view(B, L, H, Dh).transpose(1, 2)
    k = F.linear(kv_in, W[E : 2 * E], bias[E : 2 * E]).view(B, S, H, Dh).transpose(1, 2)
    v = F.linear(kv_in, W[2 * E :], bias[2 * E :]).view(B, S, H, Dh).transpose(1, 2)
    sc = torch.matmul(q * (1.0 / math.sqrt(Dh)), k.transpose(-1, -2))
    if attn_mask is not None:
        sc = sc + attn_mask
    if key_pad is not None:
        sc = sc.masked_fill(key_pad.view(B, 1, 1, S), float("-inf"))
    o = torch.matmul(F.softmax(sc, dim=-1), v).transpose(1, 2).reshape(B, L, E)
    return F.linear(o, sd[pfx + "att.out_proj.weight"], sd[pfx + "att.out_proj.bias"])


def decode(tgt: Tensor, enc_out: Tensor, enc_len: Optional[Tensor], sd: SD, cfg: ModelCfg, pfx: str = "") -> Tensor:
    """TransformerASR.decode (TransformerASR.py:426-473): tokens [n,L] -> [n,L,d].

    NormalizedEmbedding = emb*sqrt(d) (Transformer.py:966-995) + absolute PE;
    pre-norm TransformerDecoderLayer x N (Transformer.py:751-834) with GELU FFN,
    eps 1e-6 norms; final LayerNorm (Transformer.py:915-963).
    """
    n, L = tgt.shape
    d = cfg.d_model
    causal = torch.full((L, L), float("-inf")).triu(1)  # get_lookahead_mask :1037-1068
    mem_pad = None
    if enc_len is not None:
        mem_pad = ~length_to_mask(enc_len, enc_out.shape[1])
    x = F.embedding(tgt.long(), sd[pfx + "custom_tgt_module.layers.0.emb.Embedding.weight"]) * math.sqrt(d)
    x = x + abs_pos_encoding(L, d)[None]
    for l in range(cfg.num_decoder_layers):
        p = f"{pfx}decoder.layers.{l}."
        h = _ln(x, sd, p + "norm1.norm.", 1e-6)
        x = x + _mha(h, h, sd, p + "self_attn.", cfg.nhead, attn_mask=causal)
        h = _ln(x, sd, p + "norm2.norm.", 1e-6)
        x = x + _mha(h, enc_out, sd, p + "multihead_attn.", cfg.nhead, key_pad=mem_pad)
        h = _ln(x, sd, p + "norm3.norm.", 1e-6)
        x = x + _ffn(h, sd, p + "pos_ffn.", F.gelu)
    return _ln(x, sd, pfx + "decoder.norm.norm.", 1e-6)


# --------------------------------------------------------------------------
# a20: TransformerLM (full scorer of the test-time search)
# --------------------------------------------------------------------------
def lm_forward(tokens: Tensor, sd: SD, cfg: LMCfg, pfx: str = "") -> Tensor:
    """TransformerLM.forward (TransformerLM.py:116-158): tokens [n,L] -> logits [n,L,V].

    make_masks (:165-187): causal mask + key padding mask of tokens == pad_idx (0).
    NormalizedEmbedding*sqrt(d) + absolute PE; TransformerEncoderLayer x N
    (Transformer.py:427-481, post-norm unless normalize_before) with regularMHA;
    TransformerEncoder's final LayerNorm (:620-624); output_proj = Linear ->
    LayerNorm(eps 1e-6) -> Linear (TransformerLM.py:104-108).
    """
    n, L = tokens.shape
    d = cfg.d_model
    act = {"gelu": F.gelu, "relu": F.relu}[cfg.activation]
    causal = torch.full((L, L), float("-inf")).triu(1)
    key_pad = tokens.long() == cfg.pad_idx
    x = F.embedding(tokens.long(), sd[pfx + "custom_src_module.emb.Embedding.weight"]) * math.sqrt(d)
    x = x + abs_pos_encoding(L, d)[None]
    for l in range(cfg.num_encoder_layers):
        p = f"{pfx}encoder.layers.{l}."
        if cfg.normalize_before:
            h = _ln(x, sd, p + "norm1.norm.", 1e-6)
            x = x + _mha(h, h, sd, p + "self_att.", cfg.nhead, attn_mask=causal, key_pad=key_pad)
            h = _ln(x, sd, p + "norm2.norm.", 1e-6)
            x = x + _ffn(h, sd, p + "pos_ffn.", act)
        else:
            x = _ln(x + _mha(x, x, sd, p + "self_att.", cfg.nhead, attn_mask=causal, key_pad=key_pad), sd, p + "norm1.norm.", 1e-6)
            x = _ln(x + _ffn(x, sd, p + "pos_ffn.", act), sd, p + "norm2.norm.", 1e-6)
    x = _ln(x, sd, pfx + "encoder.norm.norm.", 1e-6)
    x = F.linear(x, sd[pfx + "output_proj.layers.0.w.weight"], sd[pfx + "output_proj.layers.0.w.bias"])
    x = _ln(x, sd, pfx + "output_proj.layers.1.norm.", 1e-6)
    return F.linear(x, sd[pfx + "output_proj.layers.2.w.weight"], sd[pfx + "output_proj.layers.2.w.bias"])


# --------------------------------------------------------------------------
# a14, a19: CTC prefix scorer (Watanabe et al. 2017, Alg. 2)
# --------------------------------------------------------------------------
class CTCPrefixScorer:
    """decoders/ctc.py:26-295 (full-vocabulary mode, ctc_window_size = 0).

    State per hypothesis: r [T,2,n_bh] (non-blank / blank forward log-probs of
    the prefix) and psi [n_bh,V].  Uses the finite -1e20 sentinel (:53).
    """

    NEG = -1e20

    def __init__(self, logp: Tensor, enc_lens: Tensor, blank: int, eos: int):
        # logp: [B,T,V] log-softmax of ctc_lin(enc) (scorer.py:239-255)
        x = logp.clone()
        B, T, V = x.shape
        self.B, self.T, self.V, self.blank, self.eos = B, T, V, blank, eos
        pad = ~length_to_mask(enc_lens, T)  # frames >= len (:58-61)
        x.masked_fill_(pad.unsqueeze(-1), self.NEG)
        x[:, :, 0] = x[:, :, 0].masked_fill(pad, 0.0)  # column 0, as the reference does
        self.x_nb = x.transpose(0, 1).contiguous()  # [T,B,V]
        self.x_b = self.x_nb[:, :, blank].clone()  # [T,B]
        self.last = (enc_lens - 1).long()
        self.prefix_len = -1

    def step(self, last_tok: Tensor, state, beam: int):
        """forward_step (:79-241). Returns (psi - psi_prev, (r_full, psi))."""
        T, B, V, NEG = self.T, self.B, self.V, self.NEG
        n_bh = last_tok.shape[0]
        self.prefix_len += 1
        if state is None:
            r_prev = torch.full((T, 2, B, beam), NEG)
            r_prev[:, 1] = torch.cumsum(self.x_b, 0).unsqueeze(2)
            r_prev = r_prev.view(T, 2, n_bh)
            psi_prev = torch.zeros(n_bh, V)
        else:
            r_prev, psi_prev = state
        x_nb = self.x_nb.repeat_interleave(beam, dim=1)  # [T,n_bh,V]
        x_b = self.x_b.repeat_interleave(beam, dim=1).unsqueeze(-1)  # [T,n_bh,1]
        r_sum = torch.logsumexp(r_prev, 1)  # [T,n_bh]
        phi = r_sum.unsqueeze(2).repeat(1, 1, V)
        idx = torch.arange(n_bh)
        phi[:, idx, last_tok.long()] = r_prev[:, 1, :]  # same-last-token correction (:185-186)
        start = max(1, self.prefix_len)
        r = torch.full((T, 2, n_bh, V), NEG)
        if self.prefix_len == 0:
            r[0, 0] = x_nb[0]
        for t in range(start, T):
            nb = torch.logsumexp(torch.stack([r[t - 1, 0], phi[t - 1]]), 0) + x_nb[t]
            bl = torch.logsumexp(torch.stack([r[t - 1, 0], r[t - 1, 1]]), 0) + x_b[t]
            r[t, 0], r[t, 1] = nb, bl
        psi_init = r[start - 1, 0].unsqueeze(0)
        phix = torch.cat((phi[0].unsqueeze(0), phi[:-1]), dim=0) + x_nb
        psi = torch.logsumexp(torch.cat((phix[start:T], psi_init), dim=0), dim=0)
        psi[idx, self.eos] = r_sum[self.last.repeat_interleave(beam), idx]
        if self.eos != self.blank:
            psi[:, self.blank] = NEG
        return psi - psi_prev, (r, psi)

    def permute(self, state, cand: Tensor, beam: int):
        """permute_mem (:243-295): cand [B,beam] indexes beam*V per utterance."""
        r, psi = state
        V = self.V
        n_bh = self.B * beam
        off = (torch.arange(self.B) * beam).unsqueeze(1)
        flat = (cand + off * V).view(n_bh)  # index into [n_bh*V]
        psi_sel = psi.reshape(-1)[flat].view(-1, 1).repeat(1, V)
        r_sel = r.reshape(self.T, 2, n_bh * V)[:, :, flat]
        return r_sel, psi_sel


# --------------------------------------------------------------------------
# a15, a16, a18: S2STransformerBeamSearcher
# --------------------------------------------------------------------------
@dataclass
class SearchTrace:
    """Per-step intermediates kept for parity debugging."""

    am_log_probs: List[Tensor] = field(default_factory=list)  # attn_weight * log_softmax, [n_bh,V]
    ctc_scores: List[Tensor] = field(default_factory=list)
    lm_log_probs: List[Tensor] = field(default_factory=list)
    tokens: List[Tensor] = field(default_factory=list)
    preds: List[Tensor] = field(default_factory=list)
    scores: List[Tensor] = field(default_factory=list)


def beam_search(enc: Tensor, wav_len: Tensor, sd: SD, cfg: ModelCfg, sc: SearchCfg, pfx: str = "Transformer.",
                seq_lin: str = "seq_lin.w.", ctc_lin: str = "ctc_lin.w.", trace: Optional[SearchTrace] = None,
                lm_cfg: Optional[LMCfg] = None, lm_pfx: str = "LM."):
    """S2SBeamSearcher.forward (decoders/seq2seq.py:1632-1723) specialised to
    S2STransformerBeamSearcher (:1853-1934) with an optional full CTC scorer
    (scorer.py:1221-1268).  Returns (hyps, best_lens, best_scores, best_log_probs).
    """
    B, T, _ = enc.shape
    beam, V = sc.beam, cfg.vocab
    n_bh = B * beam
    enc_lens = torch.round(T * wav_len).int()
    attn_w = 1.0 - sc.ctc_weight if sc.ctc_weight > 0 else 1.0
    ctc = None
    if sc.ctc_weight > 0:  # CTCScorer.reset_mem (scorer.py:239-255)
        logp = F.log_softmax(F.linear(enc, sd[ctc_lin + "weight"], sd[ctc_lin + "bias"]), dim=-1)
        ctc = CTCPrefixScorer(logp, enc_lens, sc.blank, sc.eos)
    ctc_state = None
    enc_i = enc.repeat_interleave(beam, dim=0)
    enc_lens_i = enc_lens.repeat_interleave(beam, dim=0)
    tok = torch.full((n_bh,), sc.bos, dtype=torch.long)
    off = torch.arange(B) * beam
    seq_scores = torch.full((n_bh,), float("-inf"))
    seq_scores[off] = 0.0
    memory = torch.empty(n_bh, 0)  # float, like decoders/utils.py:14-32
    alive_seq = torch.empty(n_bh, 0, dtype=torch.long)
    alive_lp = torch.empty(n_bh, 0)
    finished: List[list] = [[] for _ in range(B)]
    min_steps = int(T * sc.min_decode_ratio)
    max_steps = int(T * sc.max_decode_ratio)
    scores = None

    def harvest(tokens, scores_now):
        for i in torch.nonzero(tokens == sc.eos).flatten().tolist():
            b = i // beam
            if len(finished[b]) == beam:
                continue
            finished[b].append((alive_seq[i].clone(), alive_lp[i].clone(), scores_now[i].clone()))

    for step in range(max_steps):
        if all(len(f) == beam for f in finished):
            break
        memory = torch.cat([memory, tok.unsqueeze(1).to(memory.dtype)], dim=-1)
        pred = decode(memory, enc_i, enc_lens_i, sd, cfg, pfx)
        logits = F.linear(pred, sd[seq_lin + "weight"], sd[seq_lin + "bias"])
        lp = attn_w * F.log_softmax(logits / sc.temperature, dim=-1)[:, -1, :]
        lp_keep = lp.clone()
        if step < min_steps:
            lp[:, sc.eos] = sc.minus_inf
        if sc.using_eos_threshold:  # seq2seq.py:829-850,1012-1017
            mx = lp.max(dim=-1).values
            keep = lp[:, sc.eos] > sc.eos_threshold * mx
            lp[:, sc.eos] = torch.where(keep, lp[:, sc.eos], torch.full_like(mx, sc.minus_inf))
        if lm_cfg is not None and sc.lm_weight != 0.0:  # TransformerLMScorer.score (scorer.py:510-543), first full scorer
            lm_logits = lm_forward(memory, sd, lm_cfg, lm_pfx)
            lm_lp = F.log_softmax(lm_logits / sc.lm_temperature, dim=-1)[:, -1, :]
            lp = lp + lm_lp * sc.lm_weight
            if trace is not None:
                trace.lm_log_probs.append(lm_lp.clone())
        if ctc is not None:
            lp[:, sc.blank] = CTCPrefixScorer.NEG  # scorer.py:1248-1253
            ctc_sc, ctc_full = ctc.step(tok, ctc_state, beam)
            lp = lp + ctc_sc * sc.ctc_weight
            if trace is not None:
                trace.ctc_scores.append(ctc_sc.clone())
        tot = seq_scores.unsqueeze(1) + lp
        if sc.length_normalization:
            tot = tot / (step + 1)
        scores, cand = tot.view(B, -1).topk(beam, dim=-1)
        tok = (cand % V).view(n_bh)
        scores = scores.view(n_bh)
        seq_scores = scores * (step + 1) if sc.length_normalization else scores.clone()
        preds = (torch.div(cand, V, rounding_mode="floor") + off.unsqueeze(1)).view(n_bh)
        memory = memory[preds]
        if ctc is not None:
            ctc_state = ctc.permute(ctc_full, cand, beam)
        tok_lp = lp_keep.view(B, -1).gather(1, cand).view(n_bh)
        alive_seq = torch.cat([alive_seq[preds], tok.unsqueeze(1)], dim=-1)
        alive_lp = torch.cat([alive_lp[preds], tok_lp.unsqueeze(1)], dim=-1)
        if trace is not None:
            trace.am_log_probs.append(lp_keep)
            trace.tokens.append(tok.clone())
            trace.preds.append(preds.clone())
            trace.scores.append(scores.clone())
        harvest(tok, scores)
        seq_scores = seq_scores.masked_fill(tok == sc.eos, float("-inf"))

    if not all(len(f) == beam for f in finished):  # seq2seq.py:1600-1630
        harvest(torch.full((n_bh,), sc.eos, dtype=torch.long), scores)

    flat = [h for f in finished for h in f]
    max_len = max(h[0].numel() for h in flat)
    if sc.return_topk:  # _get_topk_prediction (:1418-1476): padded tensors, sequences keep their last token
        K = sc.topk
        t_hyps = torch.zeros(B, K, max_len, dtype=torch.long)
        t_lps = torch.zeros(B, K, max_len)
        t_lens, t_scores = torch.zeros(B, K), torch.zeros(B, K)
        for b in range(B):
            sc_b = torch.stack([h[2] for h in finished[b]])
            top = sc_b.topk(K)
            for r, j in enumerate(top.indices.tolist()):
                seq, lps, s = finished[b][j]
                t_hyps[b, r, : seq.numel()] = seq
                t_lps[b, r, : lps.numel()] = lps
                t_lens[b, r] = (seq.numel() - 1) / max_len
                t_scores[b, r] = s
        return t_hyps, t_lens, t_scores, t_lps
    # _get_topk_prediction (:1418-1476) with topk = 1
    hyps, lens, best_scores, best_lps = [], [], [], []
    for b in range(B):
        sc_b = torch.stack([h[2] for h in finished[b]])
        j = int(sc_b.topk(1).indices[0])
        seq, lps, s = finished[b][j]
        rel = (seq.numel() - 1) / max_len
        n_keep = int(torch.round(torch.tensor(rel * max_len)))  # undo_padding (utils/data_utils.py:28-58)
        hyps.append(seq[:n_keep].tolist())
        lens.append(rel)
        best_scores.append(s)
        best_lps.append(lps)
    return hyps, torch.tensor(lens), torch.stack(best_scores), best_lps


def greedy_search(enc: Tensor, wav_len: Tensor, sd: SD, cfg: ModelCfg, sc: SearchCfg, pfx: str = "Transformer.",
                  seq_lin: str = "seq_lin.w.", return_logits: bool = False):
    """S2SGreedySearcher.forward (seq2seq.py:176-330) + S2STransformerGreedySearcher
    (:330-367), temperature 0 (arg-max of raw logits)."""
    B, T, _ = enc.shape
    enc_lens = torch.round(T * wav_len).int()
    tok = torch.full((B,), sc.bos, dtype=torch.long)
    memory = torch.empty(B, 0)
    ended = torch.zeros(B, dtype=torch.bool)
    lps, all_logits = [], []
    for _ in range(int(T * sc.min_decode_ratio), int(T * sc.max_decode_ratio)):
        memory = torch.cat([memory, tok.unsqueeze(1).to(memory.dtype)], dim=-1)
        pred = decode(memory, enc, enc_lens, sd, cfg, pfx)
        logits = F.linear(pred, sd[seq_lin + "weight"], sd[seq_lin + "bias"])[:, -1, :]
        all_logits.append(logits)
        tok = logits.argmax(dim=-1)
        lp = F.log_softmax(logits.float(), dim=-1)
        ended = ended | (tok == sc.eos)
        lp[ended] = float("-inf")
        tok[ended] = sc.eos
        lps.append(lp)
        if ended.all():
            break
    lp = torch.stack(lps, dim=1)
    scores, preds = lp.max(dim=-1)
    m = scores == float("-inf")
    scores[m] = 0
    preds[m] = sc.eos
    L = preds.shape[1]
    hyps, rel = [], []
    for b in range(B):
        e = (preds[b] == sc.eos).nonzero()
        n = int(e[0]) if len(e) else L
        rel.append(n / L)
        hyps.append(preds[b, : int(round(n / L * L))].tolist())
    out = (hyps, torch.tensor(rel), scores, lp)
    return out + (torch.stack(all_logits, 1),) if return_logits else out


# --------------------------------------------------------------------------
# a22: whole path
# --------------------------------------------------------------------------
def encode_batch(wavs: Tensor, wav_lens: Tensor, sd: SD, fcfg: FbankCfg, mcfg: ModelCfg, norm_mean: Tensor, norm_std: Tensor) -> Tensor:
    """EncoderDecoderASR.encode_batch (inference/ASR.py:100-129):
    Fbank -> InputNormalization(global) -> CNN -> TransformerASR.encode."""
    feats = input_norm_global(fbank(wavs, fcfg), norm_mean, norm_std)
    cnn = conv_frontend(feats, sd, "CNN.")
    return encode(cnn, wav_lens, sd, mcfg, "Transformer.")
