"""Import-time stub (oracle/make_golden.py only); the hot path never calls torchaudio.  The reference builds a
`transforms.Resample` object even when the rates agree (augment/time_domain.py:559) and then never calls it
(:576-577), so a holder that refuses to resample is all that is needed."""
import types

import torch

__version__ = "2.7.1"


class _Resample(torch.nn.Module):
    def __init__(self, orig_freq=16000, new_freq=16000, *args, **kwargs):
        super().__init__()
        self.orig_freq, self.new_freq = orig_freq, new_freq

    def forward(self, x):
        raise NotImplementedError("torchaudio stub: resampling is not on the golden path")


transforms = types.SimpleNamespace(Resample=_Resample)
