"""Resample (speechbrain_amd/augment/time_domain.py): the reference delegates to torchaudio.transforms.Resample, which
is absent here -- parity with torchaudio is UNPINNED; these are the properties its published windowed-sinc algorithm
guarantees, plus transcribe-path plumbing through AudioNormalizer."""
import math

import numpy as np
import pytest
import torch

from speechbrain_amd.augment.time_domain import Resample, sinc_resample_kernel
from speechbrain_amd.inference.interfaces import AudioNormalizer


def test_identity_and_lengths():
    x = torch.randn(2, 1000)
    assert Resample(16000, 16000)(x) is x
    for orig, new in ((8000, 16000), (44100, 16000), (48000, 16000), (22050, 16000), (16000, 8000)):
        for n in (1, 777, 4410):
            y = Resample(orig, new)(torch.randn(3, n))
            g = math.gcd(orig, new)
            assert y.shape == (3, math.ceil((new // g) * n / (orig // g)))
    y3 = Resample(8000, 16000)(torch.randn(2, 500, 3))  # [batch, time, channels]
    assert y3.shape == (2, 1000, 3)


def test_kernel_shape_and_dc_gain():
    k, width = sinc_resample_kernel(3, 1)  # 48 kHz -> 16 kHz
    assert k.shape == (1, 1, 2 * width + 3) and width == math.ceil(6 * 3 / 0.99)
    for orig, new in ((1, 2), (3, 1), (441, 160)):
        k, _ = sinc_resample_kernel(orig, new)
        assert float((k.sum(-1) - 1.0).abs().max()) < 2e-2  # every output phase passes DC with ~unit gain


@pytest.mark.parametrize("orig,new,f", [(8000, 16000, 1000.0), (48000, 16000, 3000.0), (44100, 16000, 440.0)])
def test_sinusoid_below_cutoff_is_preserved(orig, new, f):
    t_in = torch.arange(orig, dtype=torch.float64) / orig          # 1 s
    y = Resample(orig, new)(torch.sin(2 * math.pi * f * t_in).float()[None])[0]
    t_out = torch.arange(y.numel(), dtype=torch.float64) / new
    ref = torch.sin(2 * math.pi * f * t_out).float()
    mid = slice(200, -200)                                             # away from the zero-padded edges
    assert float((y[mid] - ref[mid]).abs().max()) < 2e-3


def test_audio_normalizer_resamples_and_mixes():
    norm = AudioNormalizer(16000)
    stereo_8k = torch.randn(800, 2)
    out = norm(stereo_8k, 8000)
    assert out.shape == (1600,)
    assert norm(torch.randn(1600), 16000).shape == (1600,)
    assert 8000 in norm._resamplers and len(norm._resamplers) == 1   # cached per source rate
