"""speechbrain.nnet.CNN mirror: the Conv2d flavour the ConvolutionFrontEnd instantiates."""
import math

import torch


def get_padding_elem(L_in: int, stride: int, kernel_size: int, dilation: int):
    """nnet/CNN.py:1510-1536."""
    if stride > 1:
        return [math.floor(kernel_size / 2), math.floor(kernel_size / 2)]
    L_out = math.floor((L_in - dilation * (kernel_size - 1) - 1) / stride) + 1
    return [math.floor((L_in - L_out) / 2), math.floor((L_in - L_out) / 2)]


class Conv2d(torch.nn.Module):
    """Parameter holder with the reference's names (``conv.weight`` [Cout,Cin,kF,kT], ``conv.bias``).

    Only the configuration on the ASR path is executable (3x3, stride 2, "same" reflect
    padding, fused with the following LayerNorm + LeakyReLU in csrc/convfront.hip); the
    fused block is driven by lobes.models.convolution.ConvBlock.
    """

    def __init__(self, out_channels, kernel_size, input_shape=None, in_channels=None, stride=(1, 1), dilation=(1, 1),
                 padding="same", groups=1, bias=True, padding_mode="reflect", max_norm=None, swap=False,
                 skip_transpose=False, weight_norm=False, conv_init=None):
        super().__init__()
        as2 = lambda v: (v, v) if isinstance(v, int) else tuple(v)  # noqa: E731
        self.kernel_size, self.stride, self.dilation = as2(kernel_size), as2(stride), as2(dilation)
        self.padding, self.padding_mode = padding, padding_mode
        if input_shape is None and in_channels is None:
            raise ValueError("Must provide one of input_shape or in_channels")
        if in_channels is None:
            in_channels = 1 if len(input_shape) == 3 else input_shape[-1]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv = torch.nn.Conv2d(in_channels, out_channels, self.kernel_size, stride=self.stride, padding=0,
                                    dilation=self.dilation, groups=groups, bias=bias)
        if conv_init == "kaiming":
            torch.nn.init.kaiming_normal_(self.conv.weight)
        elif conv_init == "zero":
            torch.nn.init.zeros_(self.conv.weight)

    def forward(self, x):
        raise RuntimeError("Conv2d runs fused inside ConvBlock (conv + LayerNorm + LeakyReLU) on this path")
