"""speechbrain.nnet.normalization mirror (LayerNorm only)."""
import torch

from speechbrain_amd import native


class LayerNorm(torch.nn.Module):
    """nnet/normalization.py:185-242: normalises over input_size (or input_shape[2:])."""

    def __init__(self, input_size=None, input_shape=None, eps=1e-05, elementwise_affine=True):
        super().__init__()
        if not elementwise_affine:
            raise NotImplementedError("affine-free LayerNorm is not on the ASR path")
        self.eps = eps
        if input_shape is not None:
            input_size = input_shape[2:]
        self.norm = torch.nn.LayerNorm(input_size, eps=self.eps, elementwise_affine=True)

    def forward(self, x):
        return native.layernorm(x.contiguous(), self.norm.weight.reshape(-1), self.norm.bias.reshape(-1), self.eps)
