"""speechbrain.lobes.models.transformer.TransformerASR mirror (TransformerASR.py:106-675), offline path."""
from typing import Optional

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


def make_transformer_src_tgt_masks(src, tgt=None, wav_len=None, pad_idx=0, causal: bool = False,
                                   dynchunktrain_config=None):
    """TransformerASR.py:106-164 for the offline encoder: only the key-padding mask is non-trivial."""
    if causal or dynchunktrain_config is not None or tgt is not None:
        raise NotImplementedError("causal / chunked / teacher-forced masks are outside the inference path")
    src_key_padding_mask = None
    if wav_len is not None:
        abs_len = torch.round(wav_len * src.shape[1])
        # max_len = T: what the mask is combined with; equal to the reference's abs_len.max() whenever the
        # longest item fills the batch (batch_pad_right), and it keeps the encoder free of a host sync
        src_key_padding_mask = ~length_to_mask(abs_len, max_len=src.shape[1]).bool()
    return src_key_padding_mask, None, None, None


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
        """[B,T',F',C] or [B,T',F] -> [B,T',d] (TransformerASR.py:475-544)."""
        if dynchunktrain_config is not None:
            raise NotImplementedError("dynamic chunk training / streaming is outside the offline path")
        if src.dim() == 4:
            bz, t, ch1, ch2 = src.shape
            src = src.reshape(bz, t, ch1 * ch2)
        src_key_padding_mask, _, src_mask, _ = make_transformer_src_tgt_masks(src, None, wav_len, pad_idx=pad_idx,
                                                                               causal=self.causal)
        src = self.custom_src_module(src)
        # RoPEMHA rotates q/k inside the attention kernel; RelPosMHAXL takes the sinusoid table (:519-528)
        pos_embs_source = None if self.attention_type == "RoPEMHA" else self.positional_encoding(src)
        outputs = self.encoder(src=src, src_mask=src_mask, src_key_padding_mask=src_key_padding_mask,
                               pos_embs=pos_embs_source)
        if self.output_hidden_states:
            encoder_out, _, hidden_states = outputs
            return encoder_out, hidden_states
        encoder_out, _ = outputs
        return encoder_out

    def forward(self, src, tgt=None, wav_len=None, pad_idx=0):
        if tgt is not None:
            raise NotImplementedError("teacher-forced forward (training) is outside the inference path")
        return self.encode(src, wav_len, pad_idx)


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
        raise NotImplementedError("streaming (dynamic chunk) encoding is outside the offline path")

    def make_streaming_context(self, *args, **kwargs):
        raise NotImplementedError("streaming (dynamic chunk) encoding is outside the offline path")
