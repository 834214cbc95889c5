"""speechbrain.augment.time_domain mirror: ``Resample`` (augment/time_domain.py:522-577), the one class of that
module on the inference path (AudioNormalizer resamples a file whose rate differs from the model's,
dataio/preprocess.py:49-84).

The reference delegates to ``torchaudio.transforms.Resample`` (third-party, pinned 2.5.1 / 2.7.1 in uv.lock, absent
from /root/reference and from this image).  Its published algorithm (torchaudio/functional/functional.py,
``_get_sinc_resample_kernel`` / ``_apply_sinc_resample_kernel``, default ``sinc_interp_hann``,
``lowpass_filter_width=6``, ``rolloff=0.99``) is restated here:

  orig, new     <- the two rates divided by their gcd
  base          <- min(orig, new) * rolloff                      (cut-off of the anti-aliasing low-pass)
  width         <- ceil(lowpass_filter_width * orig / base)
  kernel[p][k]  <- sinc(pi*t) * cos^2(pi*t / (2*lpw)) * base/orig,  t = clamp((k - width)/orig - p/new) * base, +-lpw)
                   for output phase p = 0..new-1 and taps k = 0..2*width+orig-1
  y             <- conv1d(pad(x, (width, width + orig)), kernel, stride=orig), phases interleaved,
                   cut to ceil(new * len / orig) samples

Host arithmetic (file loading is host work in the reference as well).  Parity with torchaudio itself is
UNPINNED (the package is not available here); tests pin the properties the algorithm guarantees (identity at equal
rates, exact output length, unit DC gain, sinusoids below the cut-off preserved to 1e-3)."""
import math

import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """(kernel [new, 1, 2*width + orig] float32, width) for rates already divided by their gcd."""
    base_freq = min(orig_freq, new_freq) * rolloff
    width = math.ceil(lowpass_filter_width * orig_freq / base_freq)
    idx = torch.arange(-width, width + orig_freq, dtype=torch.float64)[None, None] / orig_freq
    t = torch.arange(0, -new_freq, -1, dtype=torch.float64)[:, None, None] / new_freq + idx
    t = (t * base_freq).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base_freq / orig_freq)
    return kernels.to(torch.float32), width


class Resample(torch.nn.Module):
    """forward(waveforms [batch, time] or [batch, time, channels]) -> the same layout at ``new_freq``."""

    def __init__(self, orig_freq=16000, new_freq=16000, lowpass_filter_width=6, rolloff=0.99):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        g = math.gcd(self.orig_freq, self.new_freq)
        self._orig, self._new = self.orig_freq // g, self.new_freq // g
        if self.orig_freq != self.new_freq:
            kernel, self._width = sinc_resample_kernel(self._orig, self._new, lowpass_filter_width, rolloff)
            self.register_buffer("kernel", kernel)

    def forward(self, waveforms):
        if self.orig_freq == self.new_freq:
            return waveforms
        if waveforms.dim() == 2:
            x, squeeze = waveforms.unsqueeze(1), True
        elif waveforms.dim() == 3:
            x, squeeze = waveforms.transpose(1, 2), False
        else:
            raise ValueError("Input must be 2 or 3 dimensions")
        B, C, n = x.shape
        x = torch.nn.functional.pad(x.reshape(B * C, 1, n).float(), (self._width, self._width + self._orig))
        y = torch.nn.functional.conv1d(x, self.kernel.to(x.device), stride=self._orig)  # [B*C, new, frames]
        y = y.transpose(1, 2).reshape(B, C, -1)[..., : math.ceil(self._new * n / self._orig)]
        return y.squeeze(1) if squeeze else y.transpose(1, 2)
