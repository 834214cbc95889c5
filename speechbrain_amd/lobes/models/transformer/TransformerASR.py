"""speechbrain.lobes.models.transformer.TransformerASR mirror (TransformerASR.py:29-726): offline, Dynamic Chunk and
streaming encoders."""
from dataclasses import dataclass
from typing import Any, Optional

import torch
import torch.nn as nn

from speechbrain_amd.lobes.models.transformer.Transformer import NormalizedEmbedding, TransformerInterface
from speechbrain_amd.nnet.activations import Swish
from speechbrain_amd.nnet.containers import ModuleList
from speechbrain_amd.nnet.linear import Linear


def length_to_mask(length, max_len=None, dtype=None, device=None):
    """dataio/dataio.py:803-848."""
    assert len(length.shape) == 1
    if max_len is None:
        max_len = length.max().long().item()
    mask = torch.arange(max_len, device=length.device, dtype=length.dtype).expand(len(length), max_len) < length.unsqueeze(1)
    return torch.as_tensor(mask, dtype=dtype or length.dtype, device=device or length.device)


@dataclass
class TransformerASRStreamingContext:
    """TransformerASR.py:29-44."""

    dynchunktrain_config: Any
    encoder_context: Any


def make_transformer_src_mask(src, causal: bool = False, dynchunktrain_config=None):
    """The [T,T] boolean Dynamic Chunk mask (True = masked) of TransformerASR.py:47-103.  The attention kernels take
    the chunk geometry directly (csrc/relpos_attn.hip key_range); this tensor exists for the reference's call
    surface and for tests."""
    if causal:
        raise NotImplementedError("causal (look-ahead) masks are a training-time feature")
    if dynchunktrain_config is None:
        return None
    T, cs = src.size(1), dynchunktrain_config.chunk_size
    idx = torch.arange(T, device=src.device)
    chunk_end = (idx // cs + 1) * cs
    mask = idx[None] >= chunk_end[:, None]
    if not dynchunktrain_config.is_infinite_left_context():
        chunk_lo = chunk_end - cs * (dynchunktrain_config.left_context_size + 1)
        mask = mask | (idx[None] < chunk_lo[:, None])
    return mask


def make_transformer_src_tgt_masks(src, tgt=None, wav_len=None, pad_idx=0, causal: bool = False,
                                   dynchunktrain_config=None):
    """TransformerASR.py:106-164 for the encoder side: key-padding mask + (optional) Dynamic Chunk mask."""
    if causal or tgt is not None:
        raise NotImplementedError("causal / teacher-forced masks are outside the inference path")
    src_key_padding_mask = None
    if wav_len is not None:
        abs_len = torch.round(wav_len * src.shape[1])
        # max_len = T: what the mask is combined with; equal to the reference's abs_len.max() whenever the
        # longest item fills the batch (batch_pad_right), and it keeps the encoder free of a host sync
        src_key_padding_mask = ~length_to_mask(abs_len, max_len=src.shape[1]).bool()
    src_mask = make_transformer_src_mask(src, causal=causal, dynchunktrain_config=dynchunktrain_config)
    return src_key_padding_mask, None, src_mask, None


class TransformerASR(TransformerInterface):
    """Same constructor and state_dict as the reference (TransformerASR.py:167-345).

    ``encode`` runs the Conformer encoder on the MI355X kernels.  ``decode`` of the reference
    (full-prefix recomputation) is replaced by the KV-cached decode step owned by the searchers in
    ``speechbrain_amd.decoders.seq2seq``; calling ``decode`` directly runs that step sequence
    over the given prefix and returns the same ``(pred, attn=None)`` tuple.
    """

    def __init__(self, tgt_vocab, input_size, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 d_ffn=2048, dropout=0.1, activation=nn.ReLU, positional_encoding="fixed_abs_sine",
                 normalize_before=False, kernel_size: Optional[int] = 31, bias: bool = True,
                 encoder_module: str = "transformer", conformer_activation=Swish, branchformer_activation=nn.GELU,
                 attention_type: str = "regularMHA", max_length: int = 2500, causal: Optional[bool] = None,
                 csgu_linear_units: int = 3072, gate_activation=nn.Identity, use_linear_after_conv: bool = False,
                 output_hidden_states=False, layerdrop_prob=0.0):
        if causal is None:
            causal = True
        super().__init__(d_model=d_model, nhead=nhead, num_encoder_layers=num_encoder_layers,
                         num_decoder_layers=num_decoder_layers, d_ffn=d_ffn, dropout=dropout, activation=activation,
                         positional_encoding=positional_encoding, normalize_before=normalize_before,
                         kernel_size=kernel_size, bias=bias, encoder_module=encoder_module,
                         conformer_activation=conformer_activation, branchformer_activation=branchformer_activation,
                         attention_type=attention_type, max_length=max_length, causal=causal,
                         csgu_linear_units=csgu_linear_units, gate_activation=gate_activation,
                         use_linear_after_conv=use_linear_after_conv, output_hidden_states=output_hidden_states,
                         layerdrop_prob=layerdrop_prob)
        self.d_model, self.nhead, self.tgt_vocab = d_model, nhead, tgt_vocab
        self.custom_src_module = ModuleList(
            Linear(input_size=input_size, n_neurons=d_model, bias=True, combine_dims=False), nn.Dropout(dropout))
        if num_decoder_layers > 0:
            self.custom_tgt_module = ModuleList(NormalizedEmbedding(d_model, tgt_vocab))
        self._init_params()

    def _init_params(self):
        for p in self.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)

    def encode(self, src, wav_len=None, pad_idx=0, dynchunktrain_config=None):
        """[B,T',F',C] or [B,T',F] -> [B,T',d] (TransformerASR.py:475-544).  ``dynchunktrain_config``: chunked
        attention + Dynamic Chunk Convolution (what a streaming-capable model is trained and evaluated with)."""
        if src.dim() == 4:
            bz, t, ch1, ch2 = src.shape
            src = src.reshape(bz, t, ch1 * ch2)
        src_key_padding_mask, _, src_mask, _ = make_transformer_src_tgt_masks(
            src, None, wav_len, pad_idx=pad_idx, causal=self.causal, dynchunktrain_config=dynchunktrain_config)
        src = self.custom_src_module(src)
        # RoPEMHA rotates q/k inside the attention kernel; RelPosMHAXL takes the sinusoid table (:519-528)
        pos_embs_source = None if self.attention_type == "RoPEMHA" else self.positional_encoding(src)
        outputs = self.encoder(src=src, src_mask=src_mask, src_key_padding_mask=src_key_padding_mask,
                               pos_embs=pos_embs_source, dynchunktrain_config=dynchunktrain_config)
        if self.output_hidden_states:
            encoder_out, _, hidden_states = outputs
            return encoder_out, hidden_states
        encoder_out, _ = outputs
        return encoder_out

    def encode_group(self, srcs, wav_lens, dynchunktrain_config=None):
        """``encode`` of several independently padded batches at once: srcs = [[B_i,T_i,F(,C)], ...], wav_lens =
        [[B_i], ...] -> [[B_i,T_i,d], ...].  The batches' rows are laid end to end so that every projection,
        feed-forward and LayerNorm of the encoder is ONE launch over all of them; each batch keeps its own padded
        length, key lengths and position table (results equal ``encode`` batch by batch)."""
        if not hasattr(self.encoder, "forward_group") or self.output_hidden_states:
            return [self.encode(s, l, dynchunktrain_config=dynchunktrain_config) for s, l in zip(srcs, wav_lens)]
        flat, segs, tables, row0, pos0 = [], [], [], 0, 0
        for src, wav_len in zip(srcs, wav_lens):
            B, T = src.shape[0], src.shape[1]
            key_len = None
            if wav_len is not None:  # make_transformer_src_tgt_masks: abs_len = round(wav_len * T)
                key_len = torch.round(wav_len * T).to(torch.int32).clamp_(max=T)
            segs.append((row0, B, T, key_len, pos0))
            flat.append(src.reshape(B * T, -1))
            if self.attention_type != "RoPEMHA":
                tables.append(self.positional_encoding.make_pe(T).reshape(2 * T - 1, -1))
            row0, pos0 = row0 + B * T, pos0 + 2 * T - 1
        x = self.custom_src_module(torch.cat(flat, dim=0))
        pos2d = torch.cat(tables, dim=0) if tables else None
        y = self.encoder.forward_group(x, pos2d, segs, dynchunktrain_config=dynchunktrain_config)
        return [y[r0: r0 + B * T].view(B, T, -1) for r0, B, T, _, _ in segs]

    def forward(self, src, tgt=None, wav_len=None, pad_idx=0):
        if tgt is not None:
            raise NotImplementedError("teacher-forced forward (training) is outside the inference path")
        return self.encode(src, wav_len, pad_idx)

    def encode_streaming(self, src, context: TransformerASRStreamingContext):
        """One chunk [B,chunk,F(,C)] -> [B,chunk,d] (TransformerASR.py:546-640); ``context`` is mutated."""
        if src.dim() == 4:
            bz, t, ch1, ch2 = src.shape
            src = src.reshape(bz, t, ch1 * ch2)
        # the relative-position table must span the cached left context as well (:611-622)
        known_left_context = context.encoder_context.layers[0].mha_left_context
        n_pos = src.shape[-2] + (0 if known_left_context is None else known_left_context.shape[-2])
        src = self.custom_src_module(src)
        pos_embs_source = None
        if self.attention_type == "RelPosMHAXL":
            pos_embs_source = self.positional_encoding.make_pe(n_pos)
        elif self.attention_type != "RoPEMHA":
            raise NotImplementedError("streaming: RelPosMHAXL and RoPEMHA encoders are implemented")
        encoder_out, _ = self.encoder.forward_streaming(src=src, pos_embs=pos_embs_source,
                                                        context=context.encoder_context)
        return encoder_out

    def make_streaming_context(self, dynchunktrain_config, encoder_kwargs={}):
        """TransformerASR.py:642-670."""
        return TransformerASRStreamingContext(
            dynchunktrain_config=dynchunktrain_config,
            encoder_context=self.encoder.make_streaming_context(dynchunktrain_config, **encoder_kwargs))


class EncoderWrapper(nn.Module):
    """TransformerASR.py:678-726: ``forward`` = ``transformer.encode`` (so the encoder can sit inside a
    ``LengthsCapableSequential``).  As in the reference, ``forward`` takes ``wav_lens`` -- not ``lengths`` --
    so a LengthsCapableSequential calls it WITHOUT lengths and the encoder then runs unmasked (:711-714)."""

    def __init__(self, transformer, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.transformer = transformer

    def forward(self, x, wav_lens=None, pad_idx=0, **kwargs):
        return self.transformer.encode(x, wav_lens, pad_idx, **kwargs)

    def forward_streaming(self, x, context):
        """TransformerASR.py:716-720."""
        return self.transformer.encode_streaming(x, context)

    def make_streaming_context(self, *args, **kwargs):
        """TransformerASR.py:722-726."""
        return self.transformer.make_streaming_context(*args, **kwargs)
