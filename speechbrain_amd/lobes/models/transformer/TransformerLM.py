"""speechbrain.lobes.models.transformer.TransformerLM mirror (TransformerLM.py:22-187).

Same constructor, attribute names and ``state_dict`` keys as the reference; ``forward`` (tokens ->
next-token logits for every position) runs the KV-cached step of csrc/search.hip
(``sbk_lm_prefix_f32``).  The same step is what ``TransformerLMScorer`` contributes to
``S2STransformerBeamSearcher`` (one position per decoding step, no prefix recomputation).
"""
import torch
from torch import nn

from speechbrain_amd import native
from speechbrain_amd.lobes.models.transformer.Transformer import NormalizedEmbedding, TransformerInterface
from speechbrain_amd.nnet.containers import ModuleList
from speechbrain_amd.nnet.linear import Linear
from speechbrain_amd.nnet.normalization import LayerNorm


class TransformerLM(TransformerInterface):
    """Encoder-only Transformer language model (TransformerLM.py:22-114)."""

    def __init__(self, vocab, d_model=512, nhead=8, num_encoder_layers=12, num_decoder_layers=0, d_ffn=2048,
                 dropout=0.1, activation=nn.ReLU, positional_encoding="fixed_abs_sine", normalize_before=False,
                 d_embedding=None, max_length=2500, causal=True, attention_type="regularMHA",
                 decoder_use_memory=False):
        if num_decoder_layers > 0 or d_embedding is not None or attention_type != "regularMHA" or not causal:
            raise NotImplementedError("TransformerLM: the scorer path implements the encoder-only, causal, regularMHA "
                                      "model without embedding_proj (the LibriSpeech LM recipe)")
        if positional_encoding != "fixed_abs_sine":
            raise NotImplementedError("TransformerLM: positional_encoding='fixed_abs_sine' only")
        super().__init__(d_model=d_model, nhead=nhead, num_encoder_layers=num_encoder_layers,
                         num_decoder_layers=num_decoder_layers, d_ffn=d_ffn, dropout=dropout, activation=activation,
                         positional_encoding=positional_encoding, normalize_before=normalize_before,
                         max_length=max_length, causal=causal, attention_type=attention_type)
        self.d_embedding = d_model
        self.custom_src_module = NormalizedEmbedding(self.d_embedding, vocab)
        self.embedding_proj = None
        self.output_proj = ModuleList(Linear(input_size=d_model, n_neurons=d_model), LayerNorm(d_model, eps=1e-6),
                                      Linear(input_size=d_model, n_neurons=vocab))
        self.num_encoder_layers = num_encoder_layers
        self.num_decoder_layers = num_decoder_layers
        self.decoder_use_memory = decoder_use_memory
        self._handle = None
        self._reset_params()

    def _reset_params(self):
        for p in self.parameters():  # TransformerLM.py:160-163
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)

    def handle(self):
        """Device-pointer view of the weights for the C ABI (rebuilt when a parameter moved or changed)."""
        if self._handle is None or self._handle.stale(self):
            self._handle = native.LMHandle(self)
        return self._handle

    def forward(self, src):
        """src [B,L] token ids -> logits [B,L,vocab] (TransformerLM.py:116-158)."""
        with torch.no_grad():
            return native.lm_prefix(self.handle(), src.to(torch.int32).contiguous())
