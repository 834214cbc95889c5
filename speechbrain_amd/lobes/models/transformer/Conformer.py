"""speechbrain.lobes.models.transformer.Conformer mirror (Conformer.py:30-848): offline, Dynamic Chunk (masked) and
streaming (chunk-by-chunk with left-context caches) paths.

ConvolutionModule / ConformerEncoderLayer / ConformerEncoder keep the reference's constructors,
forward signatures and state_dict keys.  Per layer the work is 17 HIP launches: every LayerNorm
is one row kernel, every Linear / pointwise conv one MFMA GEMM with bias + activation + (scaled)
residual fused in the epilogue, attention and GLU+depthwise-conv one fused kernel each.
"""
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn

from speechbrain_amd import native
from speechbrain_amd.nnet.activations import Swish
from speechbrain_amd.nnet.attention import PositionalwiseFeedForward, RelPosMHAXL, RoPEMHA, _act_code
from speechbrain_amd.nnet.normalization import LayerNorm


def _norm(x, ln, w, act=0):
    """act(LayerNorm(x)) for the contraction with the weight ``w`` that consumes it: written directly as that
    contraction's pre-split panel operand (native.layernorm_x3p) when it takes the both-operands-pre-split route on the
    fp32 path -- the fp32 round trip and the split pass disappear --, as an fp32 tensor otherwise."""
    if native.panel_for(x, w):
        return native.layernorm_x3p(x, ln.weight, ln.bias, ln.eps, act=act)
    return native.layernorm(x, ln.weight, ln.bias, ln.eps, act=act)


@dataclass
class ConformerEncoderLayerStreamingContext:
    """Conformer.py:30-58: per-layer state carried across chunks."""

    mha_left_context_size: int
    mha_left_context: Optional[torch.Tensor] = None      # inputs of the MHA for the last <= size frames
    dcconv_left_context: Optional[torch.Tensor] = None   # inputs of the conv module for the last `padding` frames


@dataclass
class ConformerEncoderStreamingContext:
    """Conformer.py:61-72."""

    dynchunktrain_config: object
    layers: List[ConformerEncoderLayerStreamingContext]


class ConvolutionModule(nn.Module):
    """LN -> pointwise(d->2d) -> GLU -> depthwise(k) -> LN -> act -> Linear(d->d) -> mask (Conformer.py:75-330)."""

    def __init__(self, input_size, kernel_size=31, bias=True, activation=Swish, dropout=0.0, causal=False, dilation=1):
        super().__init__()
        if causal or dilation != 1 or not bias:
            raise NotImplementedError("causal / dilated / bias-free ConvolutionModule is not on the offline ASR path")
        self.kernel_size, self.causal, self.dilation = kernel_size, causal, dilation
        self.padding = (kernel_size - 1) // 2
        self.layer_norm = nn.LayerNorm(input_size)
        self.bottleneck = nn.Sequential(nn.Conv1d(input_size, 2 * input_size, kernel_size=1, stride=1, bias=bias),
                                        nn.GLU(dim=1))
        self.conv = nn.Conv1d(input_size, input_size, kernel_size=kernel_size, stride=1, padding=self.padding,
                              dilation=dilation, groups=input_size, bias=bias)
        self.after_conv = nn.Sequential(nn.LayerNorm(input_size), activation(), nn.Linear(input_size, input_size, bias=bias),
                                        nn.Dropout(dropout))
        self.act_code = _act_code(self.after_conv[1])

    def forward(self, x, mask: Optional[torch.Tensor] = None, dynchunktrain_config=None, residual=None, key_len=None):
        """x [B,T,d]; mask [B,T,1] True = padded (reference surface) or key_len int32 [B].  With
        ``dynchunktrain_config`` the depthwise conv is the Dynamic Chunk Convolution (Conformer.py:190-313): frames
        see their past across chunk borders and zeros beyond the end of their own chunk."""
        B, T, d = x.shape
        chunk_size = int(dynchunktrain_config.chunk_size) if dynchunktrain_config is not None else 0
        if key_len is None and mask is not None:
            key_len = (~mask.reshape(B, T)).sum(-1, dtype=torch.int32)
        pw = self.bottleneck[0]
        pw_w = pw.weight.reshape(2 * d, d)
        h = _norm(x.contiguous(), self.layer_norm, pw_w)
        h = native.gemm_nt(h, pw_w, pw.bias)
        h = native.glu_dwconv(h, self.conv.weight.reshape(d, self.kernel_size), self.conv.bias, self.kernel_size,
                              chunk_size)
        lin = self.after_conv[2]
        h = _norm(h, self.after_conv[0], lin.weight, act=self.act_code)
        return native.gemm_nt(h, lin.weight, lin.bias, residual=residual, seq_len=key_len, rows_per_seq=T)

    def forward_group(self, x, segs, dynchunktrain_config=None):
        """x + module(x) over several independently padded batches laid end to end: x [M,d], segs = [(row0, B, T,
        key_len, pos0)].  LayerNorms and the bottleneck run once over all rows; the depthwise convolution and the
        length-masked output projection once per batch."""
        d = x.shape[-1]
        chunk_size = int(dynchunktrain_config.chunk_size) if dynchunktrain_config is not None else 0
        pw = self.bottleneck[0]
        pw_w = pw.weight.reshape(2 * d, d)
        h = native.gemm_nt(_norm(x, self.layer_norm, pw_w), pw_w, pw.bias)
        g = torch.empty_like(x)
        w = self.conv.weight.reshape(d, self.kernel_size)
        for row0, B, T, _, _ in segs:
            native.glu_dwconv(h[row0: row0 + B * T].view(B, T, 2 * d), w, self.conv.bias, self.kernel_size, chunk_size,
                              out=g[row0: row0 + B * T].view(B, T, d))
        ln = self.after_conv[0]
        g = native.layernorm(g, ln.weight, ln.bias, ln.eps, act=self.act_code)
        lin = self.after_conv[2]
        out = torch.empty_like(x)
        for row0, B, T, key_len, _ in segs:
            rows = slice(row0, row0 + B * T)
            native.gemm_nt(g[rows], lin.weight, lin.bias, residual=x[rows], out=out[rows], seq_len=key_len, rows_per_seq=T)
        return out


class ConformerEncoderLayer(nn.Module):
    """Conformer.py:333-499: x + FFN/2 -> MHSA -> Conv -> LN(x + FFN/2)."""

    def __init__(self, d_model, d_ffn, nhead, kernel_size=31, kdim=None, vdim=None, activation=Swish, bias=True,
                 dropout=0.0, causal=False, attention_type="RelPosMHAXL"):
        super().__init__()
        if attention_type == "RelPosMHAXL":
            self.mha_layer = RelPosMHAXL(num_heads=nhead, embed_dim=d_model, dropout=dropout, mask_pos_future=causal)
        elif attention_type == "RoPEMHA":
            self.mha_layer = RoPEMHA(num_heads=nhead, embed_dim=d_model, dropout=dropout)
        else:
            raise NotImplementedError(f"attention_type={attention_type}: RelPosMHAXL and RoPEMHA are implemented")
        self.attention_type = attention_type
        self.convolution_module = ConvolutionModule(d_model, kernel_size, bias, activation, dropout, causal=causal)
        self.ffn_module1 = nn.Sequential(
            nn.LayerNorm(d_model),
            PositionalwiseFeedForward(d_ffn=d_ffn, input_size=d_model, dropout=dropout, activation=activation),
            nn.Dropout(dropout))
        self.ffn_module2 = nn.Sequential(
            nn.LayerNorm(d_model),
            PositionalwiseFeedForward(d_ffn=d_ffn, input_size=d_model, dropout=dropout, activation=activation),
            nn.Dropout(dropout))
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.drop = nn.Dropout(dropout)
        self.collect_attention = False  # attention maps are opt-in (they are [B,H,T,T] of HBM traffic)

    def _ffn(self, mod, x):
        ln, ffn = mod[0], mod[1]
        if (native.precision() == "bf16" and native.BF16_ACTIVATIONS and x.numel() // x.shape[-1] >= native.BF16A_MIN_ROWS
                and native.bf16a_ok(x.shape[-1]) and native.bf16a_ok(ffn.ffn[0].out_features)):
            # opt-in bf16 operands: LayerNorm writes the first contraction's operand as bf16 (same rounding the fp32-A
            # kernel applies on load), the feed-forward pair keeps its hidden layer in bf16
            h = native.layernorm_bf16(x, ln.weight, ln.bias, ln.eps)
        else:
            h = _norm(x, ln, ffn.ffn[0].weight)
        return ffn(h, residual=x, alpha=0.5)

    def forward(self, x, src_mask=None, src_key_padding_mask=None, pos_embs=None, dynchunktrain_config=None,
                key_len=None):
        """``src_mask`` is taken to be make_transformer_src_mask(dynchunktrain_config) (TransformerASR.py:47-103): the
        attention kernel gets the chunk geometry instead of a [T,T] tensor; other masks are not supported."""
        if src_mask is not None and dynchunktrain_config is None:
            raise NotImplementedError("src_mask without dynchunktrain_config (causal masks) is a training-time feature")
        if key_len is None and src_key_padding_mask is not None:
            key_len = (~src_key_padding_mask).sum(-1, dtype=torch.int32)
        chunk = dynchunktrain_config.kernel_args() if dynchunktrain_config is not None else (0, -1)
        x = self._ffn(self.ffn_module1, x.contiguous())
        x, attn = self._mha(x, pos_embs, key_len, chunk)
        x = self.convolution_module(x, residual=x, key_len=key_len, dynchunktrain_config=dynchunktrain_config)
        y = self._ffn(self.ffn_module2, x)
        return native.layernorm(y, self.norm2.norm.weight, self.norm2.norm.bias, self.norm2.eps), attn

    def _mha(self, x, pos_embs, key_len, chunk=(0, -1)):
        """x + MHA(norm1(x))."""
        h = _norm(x, self.norm1.norm, self.mha_layer.in_proj_weight)
        if self.attention_type == "RoPEMHA":
            return self.mha_layer.core(h, key_len, residual=x, want_attn=self.collect_attention, chunk=chunk)
        return self.mha_layer.core(h, pos_embs.reshape(-1, x.shape[-1]), key_len, residual=x,
                                   want_attn=self.collect_attention, chunk=chunk)

    def forward_group(self, x, pos2d, segs, dynchunktrain_config=None):
        """``forward`` over several independently padded batches laid end to end (x [M,d], pos2d the batches' position
        tables end to end, segs = [(row0, B, T, key_len, pos0)]): every row-wise launch (LayerNorms, projections,
        feed-forward) covers all the batches at once -- large GEMMs whose tiles fill the chip -- and only the two
        kernels that see the time axis (attention, depthwise convolution) run per batch."""
        chunk = dynchunktrain_config.kernel_args() if dynchunktrain_config is not None else (0, -1)
        x = self._ffn(self.ffn_module1, x)
        h = _norm(x, self.norm1.norm, self.mha_layer.in_proj_weight)
        if self.attention_type == "RoPEMHA":
            x = self.mha_layer.core_group(h, segs, residual=x, chunk=chunk)
        else:
            x = self.mha_layer.core_group(h, pos2d, segs, residual=x, chunk=chunk)
        x = self.convolution_module.forward_group(x, segs, dynchunktrain_config=dynchunktrain_config)
        y = self._ffn(self.ffn_module2, x)
        return native.layernorm(y, self.norm2.norm.weight, self.norm2.norm.bias, self.norm2.eps)

    def forward_streaming(self, x, context: ConformerEncoderLayerStreamingContext, pos_embs=None):
        """One chunk through the layer with the left-context caches of ``context`` (Conformer.py:501-586): the MHA
        runs unmasked over (cached inputs | chunk), the convolution module over (cached inputs | chunk), and the
        outputs of the cached frames are dropped."""
        orig_len = x.shape[-2]
        x = self._ffn(self.ffn_module1, x.contiguous())
        if context.mha_left_context is not None:
            x = torch.cat((context.mha_left_context, x), dim=1)
        if context.mha_left_context_size > 0:
            context.mha_left_context = x[..., -context.mha_left_context_size:, :]
        x, attn = self._mha(x.contiguous(), pos_embs, None)
        x = x[..., -orig_len:, :]
        if context.dcconv_left_context is not None:
            x = torch.cat((context.dcconv_left_context, x), dim=1)
        context.dcconv_left_context = x[..., -self.convolution_module.padding:, :]
        x = x.contiguous()
        x = self.convolution_module(x, residual=x)
        x = x[..., -orig_len:, :].contiguous()
        y = self._ffn(self.ffn_module2, x)
        return native.layernorm(y, self.norm2.norm.weight, self.norm2.norm.bias, self.norm2.eps), attn

    def make_streaming_context(self, mha_left_context_size: int):
        return ConformerEncoderLayerStreamingContext(mha_left_context_size=mha_left_context_size)


class ConformerEncoder(nn.Module):
    """Conformer.py:589-778."""

    def __init__(self, num_layers, d_model, d_ffn, nhead, kernel_size=31, kdim=None, vdim=None, activation=Swish,
                 bias=True, dropout=0.0, causal=False, attention_type="RelPosMHAXL", output_hidden_states=False,
                 layerdrop_prob=0.0):
        super().__init__()
        self.layers = nn.ModuleList([
            ConformerEncoderLayer(d_ffn=d_ffn, nhead=nhead, d_model=d_model, kdim=kdim, vdim=vdim, dropout=dropout,
                                  activation=activation, kernel_size=kernel_size, bias=bias, causal=causal,
                                  attention_type=attention_type)
            for _ in range(num_layers)])
        self.norm = LayerNorm(d_model, eps=1e-6)
        self.layerdrop_prob = layerdrop_prob
        self.attention_type = attention_type
        self.output_hidden_states = output_hidden_states

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos_embs=None, dynchunktrain_config=None):
        if self.attention_type == "RelPosMHAXL" and pos_embs is None:
            raise ValueError("RelPosMHAXL needs positional embeddings")
        key_len = None
        if src_key_padding_mask is not None:
            key_len = (~src_key_padding_mask).sum(-1, dtype=torch.int32)
        output = src
        attention_lst = []
        hidden = [output] if self.output_hidden_states else None
        for layer in self.layers:
            output, attention = layer(output, src_mask=src_mask, pos_embs=pos_embs, key_len=key_len,
                                      dynchunktrain_config=dynchunktrain_config)
            attention_lst.append(attention)
            if hidden is not None:
                hidden.append(output)
        output = self.norm(output)
        if hidden is not None:
            return output, attention_lst, hidden
        return output, attention_lst

    def forward_group(self, x, pos2d, segs, dynchunktrain_config=None):
        """The layers + final norm over several independently padded batches laid end to end (see
        ConformerEncoderLayer.forward_group); returns [M,d]."""
        if self.attention_type == "RelPosMHAXL" and pos2d is None:
            raise ValueError("RelPosMHAXL needs positional embeddings")
        for layer in self.layers:
            x = layer.forward_group(x, pos2d, segs, dynchunktrain_config=dynchunktrain_config)
        return self.norm(x)

    def forward_streaming(self, src, context: ConformerEncoderStreamingContext, pos_embs=None):
        """Conformer.py:780-828."""
        if self.attention_type == "RelPosMHAXL" and pos_embs is None:
            raise ValueError("RelPosMHAXL needs positional embeddings")
        output, attention_lst = src, []
        for i, layer in enumerate(self.layers):
            output, attention = layer.forward_streaming(output, pos_embs=pos_embs, context=context.layers[i])
            attention_lst.append(attention)
        return self.norm(output), attention_lst

    def make_streaming_context(self, dynchunktrain_config):
        """Conformer.py:830-848."""
        return ConformerEncoderStreamingContext(
            dynchunktrain_config=dynchunktrain_config,
            layers=[layer.make_streaming_context(mha_left_context_size=dynchunktrain_config.left_context_size_frames())
                    for layer in self.layers])
