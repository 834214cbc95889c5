"""speechbrain.nnet.linear mirror."""
import torch

from speechbrain_amd import native


class Linear(torch.nn.Module):
    """nnet/linear.py:16-91: y = x W^T + b on the last dimension (fp32 MFMA GEMM)."""

    def __init__(self, n_neurons, input_shape=None, input_size=None, bias=True, max_norm=None, combine_dims=False):
        super().__init__()
        if max_norm is not None:
            raise NotImplementedError("max_norm is a training-time constraint")
        self.combine_dims = combine_dims
        if input_shape is None and input_size is None:
            raise ValueError("Expected one of input_shape or input_size")
        if input_size is None:
            input_size = input_shape[-1]
            if len(input_shape) == 4 and combine_dims:
                input_size = input_shape[2] * input_shape[3]
        self.w = torch.nn.Linear(input_size, n_neurons, bias=bias)

    def forward(self, x):
        if x.ndim == 4 and self.combine_dims:
            x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
        return native.gemm_nt(x.contiguous(), self.w.weight, self.w.bias)
