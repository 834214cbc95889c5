"""speechbrain.nnet.activations mirror: only what the Conformer path instantiates."""
import torch


class Swish(torch.nn.Module):
    """x * sigmoid(beta * x) (nnet/activations.py:133-171).  On the MI355X path the
    activation is fused into the producing kernel's epilogue; this module is the
    constructor-compatible marker the lobes inspect."""

    def __init__(self, beta: float = 1.0):
        super().__init__()
        self.beta = beta
        if beta != 1.0:
            raise NotImplementedError("Swish beta != 1 is not on the ASR path")

    def forward(self, x):
        raise RuntimeError("Swish is fused into the HIP epilogues; it is never called stand-alone on this path")
