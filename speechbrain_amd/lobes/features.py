"""speechbrain.lobes.features mirror: Fbank (lobes/features.py:38-173)."""
import torch

from speechbrain_amd.processing.features import STFT, FbankFrontend, Filterbank, spectral_magnitude


class Fbank(torch.nn.Module):
    """waveform [B,N] -> log-mel filterbank [B,T,n_mels], always fp32 (the reference forces fp32
    through utils/autocast.py:167).  Same constructor as the reference; deltas / context windows /
    learnable or non-triangular filters are not on the Conformer ASR path."""

    def __init__(self, deltas=False, context=False, requires_grad=False, sample_rate=16000, f_min=0, f_max=None,
                 n_fft=400, n_mels=40, filter_shape="triangular", param_change_factor=1.0, param_rand_factor=0.0,
                 left_frames=5, right_frames=5, win_length=25, hop_length=10):
        super().__init__()
        if deltas or context or requires_grad or filter_shape != "triangular" or param_rand_factor != 0.0:
            raise NotImplementedError("only plain frozen triangular Fbank is on the MI355X ASR path")
        self.deltas, self.context, self.requires_grad = deltas, context, requires_grad
        if f_max is None:
            f_max = sample_rate / 2
        # the reference's stages, available stand-alone (same attribute names: lobes/features.py:117-145) ...
        self.compute_STFT = STFT(sample_rate=sample_rate, n_fft=n_fft, win_length=win_length, hop_length=hop_length)
        self.compute_fbanks = Filterbank(sample_rate=sample_rate, n_fft=n_fft, n_mels=n_mels, f_min=f_min,
                                         f_max=f_max, freeze=not requires_grad, filter_shape=filter_shape,
                                         param_change_factor=param_change_factor, param_rand_factor=param_rand_factor)
        # ... and the fused single-pass kernel that forward() uses
        self.fused = FbankFrontend(sample_rate=sample_rate, win_length=win_length, hop_length=hop_length,
                                   n_fft=n_fft, n_mels=n_mels, f_min=f_min, f_max=f_max)

    def forward(self, wav):
        return self.fused(wav)

    def forward_staged(self, wav):
        """STFT -> spectral_magnitude -> Filterbank exactly as the reference composes them (:147-169)."""
        return self.compute_fbanks(spectral_magnitude(self.compute_STFT(wav)))
