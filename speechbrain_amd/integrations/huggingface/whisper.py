"""speechbrain.integrations.huggingface.whisper mirror -- the log-mel front-end of WhisperASR (BASELINE.json
configs[4]): ``pad_or_trim`` + ``log_mel_spectrogram`` (integrations/huggingface/whisper.py:276-350), on the HIP
kernels (csrc/fbank.hip, ``sbk_whisper_log_mel_f32``).

The encoder / decoder of the reference live in HuggingFace ``transformers`` (``WhisperModel``, call sites :372-374,
:417-423; third-party, pinned 4.46.3 / 4.53.2 in uv.lock); they are NOT re-implemented here yet -- only the
front-end, which IS in-tree in the reference, is.  The mel filters are those of the HF feature extractor
(``transformers.audio_utils.mel_filter_bank(..., norm="slaney", mel_scale="slaney")``), restated below and pinned
against the installed ``transformers`` in tests/test_whisper.py."""
import math

import numpy as np
import torch

from speechbrain_amd import native
from speechbrain_amd.processing.features import factor_radices

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE  # 480000 samples in a 30-second chunk (whisper.py:25-28)


def _hz_to_mel_slaney(hz):
    hz = np.asarray(hz, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    logstep = 27.0 / np.log(6.4)
    mel = 3.0 * hz / 200.0
    return np.where(hz >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(hz, 1e-30) / min_log_hz) * logstep, mel)


def _mel_to_hz_slaney(mel):
    mel = np.asarray(mel, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(mel >= min_log_mel, min_log_hz * np.exp(logstep * (mel - min_log_mel)), 200.0 * mel / 3.0)


def slaney_mel_filters(n_mels=80, n_fft=N_FFT, sample_rate=SAMPLE_RATE, f_min=0.0, f_max=8000.0):
    """[n_fft/2+1, n_mels] triangular filters on the Slaney mel scale with Slaney (area) normalisation: what
    WhisperFeatureExtractor builds (and librosa.filters.mel does)."""
    n_bins = n_fft // 2 + 1
    fft_freqs = np.linspace(0, sample_rate // 2, n_bins)
    filter_freqs = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2))
    diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))
    fb *= (2.0 / (filter_freqs[2: n_mels + 2] - filter_freqs[:n_mels]))[None, :]
    return torch.from_numpy(fb.astype(np.float32))


class WhisperLogMel(torch.nn.Module):
    """``_get_mel`` of the reference's Whisper wrapper (whisper.py:262-316): waveforms [B, time] at 16 kHz ->
    [B, n_mels, 3000].  ``n_mels`` = 80 (up to large-v2) or 128 (large-v3)."""

    def __init__(self, n_mels=128, n_fft=N_FFT, hop_length=HOP_LENGTH, n_samples=N_SAMPLES, mel_filters=None):
        super().__init__()
        self._n_fft, self._hop_length, self._n_samples, self.n_mels = n_fft, hop_length, n_samples, n_mels
        fb = slaney_mel_filters(n_mels, n_fft) if mel_filters is None else torch.as_tensor(mel_filters, dtype=torch.float32)
        if fb.shape[0] == n_mels and fb.shape[1] != n_mels:  # the reference accepts either orientation (:176-182)
            fb = fb.t()
        w, ptr, first = [], [0], []
        for j in range(n_mels):  # CSR by filter: each triangular filter is one contiguous run of bins
            nz = torch.nonzero(fb[:, j]).flatten()
            lo, hi = (int(nz[0]), int(nz[-1]) + 1) if nz.numel() else (0, 0)
            first.append(lo)
            w.append(fb[lo:hi, j])
            ptr.append(ptr[-1] + hi - lo)
        m = torch.arange(n_fft, dtype=torch.float64) * (2.0 * math.pi / n_fft)
        self.radices = factor_radices(n_fft)
        self.register_buffer("_mel_filters", fb.t().contiguous(), persistent=False)  # [n_mels, 201] like the reference
        self.register_buffer("window", torch.hann_window(n_fft), persistent=False)
        self.register_buffer("twiddle", torch.stack([torch.cos(m), -torch.sin(m)], dim=1).float().contiguous(), persistent=False)
        self.register_buffer("mel_w", torch.cat(w).contiguous(), persistent=False)
        self.register_buffer("mel_ptr", torch.tensor(ptr, dtype=torch.int32), persistent=False)
        self.register_buffer("mel_bin", torch.tensor(first, dtype=torch.int32), persistent=False)

    def pad_or_trim(self, array, length=None, axis=-1):
        """whisper.py:318-350: zero right-pad or cut to ``length`` samples along ``axis``."""
        length = self._n_samples if length is None else length
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            pad_widths = [(0, 0)] * array.ndim
            pad_widths[axis] = (0, length - array.shape[axis])
            array = torch.nn.functional.pad(array, [p for sizes in pad_widths[::-1] for p in sizes])
        return array

    def log_mel_spectrogram(self, audio, padding: int = 0):
        if padding > 0:
            audio = torch.nn.functional.pad(audio, (0, padding))
        return native.whisper_log_mel(audio.float().contiguous(), self.window, self.twiddle, self.radices, self.mel_w,
                                      self.mel_ptr, self.mel_bin, self._n_fft, self._hop_length, self.n_mels)

    def forward(self, wav):
        return self.log_mel_spectrogram(self.pad_or_trim(wav))
