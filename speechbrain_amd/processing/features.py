"""Signal features on MI355X: host-side mirror of ``speechbrain.processing.features``.

Same class names, constructor arguments and forward semantics as the reference
(processing/features.py); the arithmetic runs in the HIP kernels of
``csrc/fbank.hip`` / ``csrc/norm.hip``.  Only what the EncoderDecoderASR path
uses is provided (STFT+power+mel+dB fused behind :class:`FbankFrontend`,
:class:`InputNormalization` in eval mode, :func:`make_padding_mask`).
"""

from __future__ import annotations

import math
from typing import List, Optional

import torch

from speechbrain_amd import native


def factor_radices(n: int) -> List[int]:
    """Factor n_fft into FFT passes of radix 4/2/3/5 (LDS Stockham FFT, csrc/fbank.hip)."""
    out, m = [], n
    for r in (4, 2, 3, 5):
        while m % r == 0:
            out.append(r)
            m //= r
    if m != 1:
        raise ValueError(f"n_fft={n} has a prime factor > 5; unsupported by the in-LDS FFT")
    return out


def make_padding_mask(x, lengths=None, length_dim=1, eps=1e-6):
    """processing/features.py:1554-1615: boolean mask, True = valid position."""
    if lengths is None:
        lengths = torch.ones(x.size(0), device=x.device)
    max_len = x.size(length_dim)
    abs_lengths = (lengths * max_len - eps).unsqueeze(1)
    mask = torch.arange(max_len, device=x.device).unsqueeze(0) < abs_lengths
    for dim in range(1, x.ndim):
        if dim != length_dim:
            mask = mask.unsqueeze(dim)
    return mask


class STFT(torch.nn.Module):
    """processing/features.py:58-188: [B,N] -> [B,T,n_fft/2+1,2] (re, im).  Same constructor as the
    reference; the periodic Hamming window / centred constant padding / one-sided unnormalised
    transform used by the ASR recipes run in the in-LDS FFT kernel (csrc/fbank.hip)."""

    def __init__(self, sample_rate, win_length=25, hop_length=10, n_fft=400, window_fn=torch.hamming_window,
                 normalized_stft=False, center=True, pad_mode="constant", onesided=True):
        super().__init__()
        if normalized_stft or not center or pad_mode != "constant" or not onesided:
            raise NotImplementedError("only the recipe configuration (center, constant pad, onesided) is implemented")
        self.sample_rate, self.n_fft = sample_rate, n_fft
        self.win_length = int(round((sample_rate / 1000.0) * win_length))
        self.hop_length = int(round((sample_rate / 1000.0) * hop_length))
        self.normalized_stft, self.center, self.pad_mode, self.onesided = normalized_stft, center, pad_mode, onesided
        window = window_fn(self.win_length)
        self.register_buffer("window", window, persistent=False)
        padded = window
        if self.win_length < n_fft:
            left = (n_fft - self.win_length) // 2
            padded = torch.nn.functional.pad(window, (left, n_fft - self.win_length - left))
        m = torch.arange(n_fft, dtype=torch.float64) * (2.0 * math.pi / n_fft)
        self.register_buffer("_window_fft", padded.float().contiguous(), persistent=False)
        self.register_buffer("_twiddle", torch.stack([torch.cos(m), -torch.sin(m)], dim=1).float().contiguous(),
                             persistent=False)
        self.radices = factor_radices(n_fft)

    def get_filter_properties(self):
        """processing/features.py:190-199."""
        from speechbrain_amd.utils.filter_analysis import FilterProperties

        if not self.center:
            raise ValueError("ValueProperties cannot model a non-centered STFT, as it assumes either centering or "
                             "causality")
        return FilterProperties(window_size=self.win_length, stride=self.hop_length)

    def forward(self, x):
        if x.dim() != 2:
            raise NotImplementedError("multi-channel input is not on the ASR path")
        return native.stft(x.float().contiguous(), self._window_fft, self._twiddle, self.radices, self.n_fft,
                           self.hop_length)


def spectral_magnitude(stft, power: float = 1, log: bool = False, eps: float = 1e-14):
    """processing/features.py:341-378: (re^2 + im^2)^power (+ optional log)."""
    return native.spectral_magnitude(stft.contiguous(), power, log, eps)


class Filterbank(torch.nn.Module):
    """processing/features.py:381-759, frozen triangular filters: spectrogram [B,T,n_stft] -> fbanks
    [B,T,n_mels] = dB(spectrogram @ filter matrix) with the per-utterance top_db floor."""

    def __init__(self, n_mels=40, log_mel=True, filter_shape="triangular", f_min=0, f_max=8000, n_fft=400,
                 sample_rate=16000, power_spectrogram=2, amin=1e-10, ref_value=1.0, top_db=80.0,
                 param_change_factor=1.0, param_rand_factor=0.0, freeze=True):
        super().__init__()
        if filter_shape != "triangular" or not freeze or param_rand_factor != 0.0:
            raise NotImplementedError("only frozen triangular filters are on the ASR path")
        self.n_mels, self.log_mel, self.n_fft, self.sample_rate = n_mels, log_mel, n_fft, sample_rate
        self.amin, self.ref_value, self.top_db = amin, ref_value, top_db
        self.n_stft = n_fft // 2 + 1
        self.db_multiplier = math.log10(max(amin, ref_value))
        self.multiplier = 10 if power_spectrogram == 2 else 20
        fb = FbankFrontend._filter_matrix(n_fft, n_mels, f_min, f_max, sample_rate)
        self.register_buffer("fbank_matrix_t", fb.t().contiguous(), persistent=False)  # [n_mels, n_stft]: GEMM weight

    def forward(self, spectrogram):
        if spectrogram.dim() != 3:
            raise NotImplementedError("multi-channel spectrograms are not on the ASR path")
        fbanks = native.gemm_nt(spectrogram.contiguous(), self.fbank_matrix_t)
        if self.log_mel:
            fbanks = native.amplitude_to_db(fbanks, self.multiplier, self.amin, self.multiplier * self.db_multiplier,
                                            self.top_db)
        return fbanks


class FbankFrontend(torch.nn.Module):
    """STFT -> |.|^2 -> triangular mel filterbank -> dB with top_db floor, one fused launch pair.

    Holds what STFT (processing/features.py:109-139) and Filterbank (:437-510,
    :620-647, frozen triangular filters) hold in the reference: the periodic
    Hamming window and the filter matrix, the latter compacted to one
    contiguous run of FFT bins per filter for LDS staging.
    """

    def __init__(self, sample_rate=16000, win_length=25, hop_length=10, n_fft=400, n_mels=40, f_min=0,
                 f_max=None, amin=1e-10, ref_value=1.0, top_db=80.0):
        super().__init__()
        if f_max is None:
            f_max = sample_rate / 2
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.win = int(round((sample_rate / 1000.0) * win_length))
        self.hop = int(round((sample_rate / 1000.0) * hop_length))
        self.amin, self.top_db = amin, top_db
        if ref_value != 1.0:
            raise NotImplementedError("ref_value != 1 (the reference default) is not on the ASR path")
        if self.win > n_fft:
            raise ValueError("win_length longer than n_fft")
        window = torch.hamming_window(self.win)
        if self.win < n_fft:  # torch.stft centres a short window inside n_fft
            left = (n_fft - self.win) // 2
            window = torch.nn.functional.pad(window, (left, n_fft - self.win - left))
        m = torch.arange(n_fft, dtype=torch.float64) * (2.0 * math.pi / n_fft)
        twiddle = torch.stack([torch.cos(m), -torch.sin(m)], dim=1).float()
        fb = self._filter_matrix(n_fft, n_mels, f_min, f_max, sample_rate)  # [n_stft, n_mels]
        w, ptr, first = [], [0], []
        for j in range(n_mels):
            nz = torch.nonzero(fb[:, j]).flatten()
            if nz.numel() == 0:
                first.append(0)
            else:
                lo, hi = int(nz[0]), int(nz[-1]) + 1
                first.append(lo)
                w.append(fb[lo:hi, j])
            ptr.append(ptr[-1] + (0 if nz.numel() == 0 else hi - lo))
        self.radices = factor_radices(n_fft)
        self.register_buffer("window", window.contiguous(), persistent=False)
        self.register_buffer("twiddle", twiddle.contiguous(), persistent=False)
        self.register_buffer("fbank_matrix", fb.contiguous(), persistent=False)
        self.register_buffer("mel_w", torch.cat(w).contiguous() if w else torch.zeros(1), persistent=False)
        self.register_buffer("mel_ptr", torch.tensor(ptr, dtype=torch.int32), persistent=False)
        self.register_buffer("mel_bin", torch.tensor(first, dtype=torch.int32), persistent=False)

    @staticmethod
    def _filter_matrix(n_fft, n_mels, f_min, f_max, sample_rate):
        # processing/features.py:487-510 and :620-647, same fp32 operation order
        n_stft = n_fft // 2 + 1
        to_mel = lambda hz: 2595 * math.log10(1 + hz / 700)  # noqa: E731
        mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
        hz = 700 * (10 ** (mel / 2595) - 1)
        band = (hz[1:] - hz[:-1])[:-1]
        f_central = hz[1:-1]
        all_freqs = torch.linspace(0, sample_rate // 2, n_stft).repeat(n_mels, 1)
        slope = (all_freqs - f_central.unsqueeze(1)) / band.unsqueeze(1)
        return torch.max(torch.zeros(1), torch.min(slope + 1.0, -slope + 1.0)).transpose(0, 1)

    def forward(self, wav, norm_mean=None, norm_std=None, norm_eps=1e-10):
        if wav.dim() != 2:
            raise NotImplementedError("multi-channel input is not on the ASR path")
        return native.fbank(wav.float().contiguous(), self.window, self.twiddle, self.radices, self.mel_w,
                            self.mel_ptr, self.mel_bin, self.n_fft, self.hop, self.n_mels, self.amin, self.top_db,
                            norm_mean, norm_std, norm_eps)


class InputNormalization(torch.nn.Module):
    """Inference-time mirror of processing/features.py:1265-1551.

    Same constructor and ``forward(x, lengths=None, epoch=None)``; statistics are
    the reference's attributes ``glob_mean`` / ``glob_std`` / ``count`` with the
    same ``_save`` / ``_load`` dictionary format.  Training-time statistic
    updates (and their all-reduce) are outside the inference path.
    """

    NORM_TYPES = ("global", "batch", "sentence")

    def __init__(self, mean_norm=True, std_norm=True, norm_type="global", avg_factor=None, length_dim=1,
                 update_until_epoch=2, avoid_padding_norm=False, epsilon=1e-10, device="cpu"):
        super().__init__()
        if not mean_norm:
            raise ValueError("Passing `False` for `mean_norm` is deprecated.")
        if norm_type not in self.NORM_TYPES:
            raise ValueError(f"norm_type must be one of {self.NORM_TYPES}.")
        self.std_norm, self.norm_type = std_norm, norm_type
        self.avoid_padding_norm, self.epsilon, self.length_dim = avoid_padding_norm, epsilon, length_dim
        self.update_until_epoch = update_until_epoch or math.inf
        self.glob_mean = torch.empty(0)
        self.glob_std = torch.empty(0)
        self.count = 0

    def forward(self, x, lengths=None, epoch=None):
        if self.norm_type in ("sentence", "batch"):  # statistics of the input itself: the same in train and eval
            if x.dim() != 3 or self.length_dim != 1:
                raise NotImplementedError("sentence / batch normalisation of [batch, time, channel] inputs is implemented")
            T = x.shape[1]
            if lengths is None:
                n_valid = torch.full((x.shape[0],), T, dtype=torch.int32, device=x.device)
            else:  # make_padding_mask (features.py:1603-1606): frame t is valid iff t < lengths * T - 1e-6
                n_valid = torch.ceil(lengths.to(x.device) * T - 1e-6).clamp_(0, T).to(torch.int32)
            return native.input_norm_stats(x.contiguous(), n_valid, self.norm_type == "batch", self.std_norm,
                                           self.epsilon, self.avoid_padding_norm)
        if self.training:
            raise NotImplementedError("statistic updates (training) are outside the MI355X inference path")
        if self.glob_mean.numel() == 0:
            raise RuntimeError("InputNormalization has no statistics loaded (glob_mean/glob_std)")
        # (locals: several worker threads may get here at once; none must see one statistic moved and the other not)
        mean, std = self.glob_mean, self.glob_std
        if mean.device != x.device or std.device != x.device:
            mean, std = mean.to(x.device), std.to(x.device)
            self.glob_mean, self.glob_std = mean, std
        if not self.std_norm:
            std = torch.ones_like(mean)
        if self.avoid_padding_norm and lengths is not None:  # padded frames: mean 0, std 1 (features.py:1447-1449)
            if x.dim() != 3 or self.length_dim != 1:
                raise NotImplementedError("avoid_padding_norm: [batch, time, channel] inputs are implemented")
            T = x.shape[1]
            n_valid = torch.ceil(lengths.to(x.device) * T - 1e-6).clamp_(0, T).to(torch.int32)
            return native.input_norm_global_masked(x.contiguous(), mean.float().contiguous(), std.float().contiguous(),
                                                   n_valid, self.epsilon)
        return native.input_norm_global(x.contiguous(), mean.float().contiguous(), std.float().contiguous(), self.epsilon)

    def _statistics_dict(self):
        return {"count": self.count, "glob_mean": self.glob_mean, "glob_std": self.glob_std}

    def _load_statistics_dict(self, state):
        self.count, self.glob_mean, self.glob_std = state["count"], state["glob_mean"], state["glob_std"]

    def to(self, device):
        self = super().to(device)
        self.glob_mean = self.glob_mean.to(device)
        self.glob_std = self.glob_std.to(device)
        return self

    def _apply(self, fn, *args, **kwargs):
        """The statistics are plain attributes (as in the reference, whose own ``to`` moves them): a parent module's
        ``.to(device)`` reaches this module only through ``_apply``, so move them here too -- they are then in
        place before any worker thread runs ``forward``."""
        super()._apply(fn, *args, **kwargs)
        if self.glob_mean.numel():
            self.glob_mean, self.glob_std = fn(self.glob_mean), fn(self.glob_std)
        return self

    def _save(self, path):
        torch.save(self._statistics_dict(), path)

    def _load(self, path, end_of_epoch=False):
        self._load_statistics_dict(torch.load(path, map_location="cpu"))
