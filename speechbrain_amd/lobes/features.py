"""speechbrain.lobes.features mirror: Fbank (lobes/features.py:38-173) and the streaming feature wrapper
(:483-670)."""
from dataclasses import dataclass
from typing import Optional

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

    def get_filter_properties(self):
        """Only the STFT shapes the time footprint of the Fbank (:171-173)."""
        return self.compute_STFT.get_filter_properties()

    def forward_staged(self, wav):
        """STFT -> spectral_magnitude -> Filterbank exactly as the reference composes them (:147-169)."""
        return self.compute_fbanks(spectral_magnitude(self.compute_STFT(wav)))


def upalign_value(x, to: int) -> int:
    """Round x up to the next multiple of `to` (:483-489)."""
    assert x >= 0
    return x if x % to == 0 else x + to - (x % to)


@dataclass
class StreamingFeatureWrapperContext:
    """Cached input frames that become the left padding of the next chunk (:492-502)."""

    left_context: Optional[torch.Tensor]


class StreamingFeatureWrapper(torch.nn.Module):
    """Run a feature pipeline chunk by chunk (:505-670): the first chunk is left-padded with two paddings' worth of
    zeros, every chunk is prefixed with the cached tail of its predecessor, and the output frames that belong to the
    padding on either side are dropped -- so every call consumes and produces the same number of frames.

    Host logic around ``module`` (e.g. Fbank -> InputNormalization -> ConvolutionFrontEnd on the HIP kernels)."""

    def __init__(self, module: torch.nn.Module, properties):
        super().__init__()
        self.module, self.properties = module, properties
        if properties.causal:
            raise ValueError("Causal streaming feature wrapper is not yet supported")
        if properties.dilation != 1:
            raise ValueError("Dilation not yet supported in streaming feature wrapper")

    def get_required_padding(self) -> int:
        return upalign_value((self.properties.window_size - 1) // 2, self.properties.stride)

    def get_output_count_per_pad_frame(self) -> int:
        return self.get_required_padding() // self.properties.stride

    def get_recommended_final_chunk_count(self, frames_per_chunk: int) -> int:
        return upalign_value(self.get_required_padding(), frames_per_chunk) // frames_per_chunk

    def forward(self, chunk, context: StreamingFeatureWrapperContext, *extra_args, **extra_kwargs):
        pad, n_drop = self.get_required_padding(), self.get_output_count_per_pad_frame()
        if context.left_context is None:
            chunk = torch.nn.functional.pad(chunk, (pad * 2, 0))
        else:
            chunk = torch.cat((context.left_context, chunk), 1)
        context.left_context = chunk[:, -pad * 2:]
        feats = self.module(chunk.contiguous(), *extra_args, **extra_kwargs)
        return feats[:, n_drop:-n_drop, ...]

    def get_filter_properties(self):
        return self.properties

    def make_streaming_context(self) -> StreamingFeatureWrapperContext:
        return StreamingFeatureWrapperContext(None)
