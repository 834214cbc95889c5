"""speechbrain.inference.interfaces mirror: the slice of ``Pretrained`` that EncoderDecoderASR needs
(inference/interfaces.py:216-489).  Construction from modules=/hparams= dicts and ``from_hparams`` on a LOCAL
model directory (hyperparams.yaml + *.ckpt, the layout of the reference's HuggingFace model cards) are
supported; fetching from HuggingFace / URLs is not (no network on the box)."""
import os
import struct
import types

import numpy as np
import torch


def read_wav(path):
    """RIFF/WAVE reader for the formats libsndfile (the reference's ``soundfile`` backend,
    dataio/audio_io.py:141-209) decodes to float32 the same way: integer PCM of 8 / 16 / 24 / 32 bits scaled by
    2^(bits-1) (8-bit is unsigned with a 128 offset), IEEE float 32 / 64 as is; plain and WAVE_FORMAT_EXTENSIBLE
    headers.  Returns (float32 array [frames, channels], sample_rate)."""
    with open(str(path), "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (only wav files are read by the built-in loader)")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            code, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if code == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the real code leads the sub-format GUID
                code = struct.unpack("<H", body[24:26])[0]
            fmt = (code, ch, sr, bits)
        elif tag == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, ch, sr, bits = fmt
    if code == 1 and bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif code == 1 and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif code == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif code == 1 and bits == 32:
        x = (np.frombuffer(pcm, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif code == 3 and bits in (32, 64):
        x = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise NotImplementedError(f"{path}: wav format code {code} with {bits} bits is not read by the built-in loader")
    n = x.size // ch * ch
    return x[:n].reshape(-1, ch), sr


class AudioNormalizer:
    """dataio/preprocess.py:8-84: resample to the model's rate (identity when the rates agree,
    augment/time_domain.py:576-577; otherwise the windowed-sinc resampler of ``augment.time_domain.Resample``, one
    cached per source rate like the reference's ``_cached_resample``) and average the channels."""

    def __init__(self, sample_rate=16000, mix="avg-to-mono"):
        if mix not in ("avg-to-mono", "keep"):
            raise ValueError(f"Unexpected mixing configuration {mix}")
        self.sample_rate, self.mix = sample_rate, mix
        self._resamplers = {}

    def __call__(self, audio, sample_rate):
        """audio [time] or [time, channels] -> [time] (mono) at ``self.sample_rate``."""
        if sample_rate != self.sample_rate:
            from speechbrain_amd.augment.time_domain import Resample

            if sample_rate not in self._resamplers:
                self._resamplers[sample_rate] = Resample(sample_rate, self.sample_rate)
            audio = self._resamplers[sample_rate](audio.unsqueeze(0)).squeeze(0)
        if audio.dim() == 2 and self.mix == "avg-to-mono":
            audio = audio.mean(dim=1)
        return audio


class Pretrained(torch.nn.Module):
    HPARAMS_NEEDED = []
    MODULES_NEEDED = []

    def __init__(self, modules=None, hparams=None, run_opts=None, freeze_params=True):
        super().__init__()
        run_opts = dict(run_opts or {})
        device = run_opts.get("device")
        if device is None:
            device = "cuda:0" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        # run_opts precision / eval_precision (utils/run_opts.py:114-115; the reference turns it into the autocast
        # context every forward runs in, inference/interfaces.py:295-298): "fp32" = the parity path; "bf16" / "fp16" =
        # operands of that type with fp32 accumulation in the encoder's large contractions (Fbank and everything else
        # stay fp32, as the reference's autocast keeps them, utils/autocast.py:167); "fp8" = the e4m3 activation pipeline
        # of the Whisper encoder (BASELINE configs[4]; native.precision_scope).  Every forward of an interface runs inside
        # native.precision_scope(self.eval_precision).
        self.eval_precision = run_opts.get("eval_precision") or run_opts.get("precision") or "fp32"
        if self.eval_precision not in ("fp32", "bf16", "fp16", "fp8"):
            raise NotImplementedError(f"precision {self.eval_precision!r}: 'fp32' (parity), 'bf16', 'fp16' and 'fp8' are implemented")
        self.mods = torch.nn.ModuleDict(modules or {})
        for m in self.mods.values():
            if m is not None:
                m.to(self.device)
        for name in self.MODULES_NEEDED:
            if name not in self.mods:
                raise ValueError(f"Need modules['{name}']")
        if self.HPARAMS_NEEDED and hparams is None:
            raise ValueError("Need to provide hparams dict.")
        if hparams is not None:
            for name in self.HPARAMS_NEEDED:
                if name not in hparams:
                    raise ValueError(f"Need hparams['{name}']")
            self.hparams = types.SimpleNamespace(**hparams)
        self.audio_normalizer = (hparams or {}).get("audio_normalizer", AudioNormalizer())
        if freeze_params:
            self.mods.eval()
            for p in self.mods.parameters():
                p.requires_grad = False

    def load_audio(self, path, savedir=None):
        """wav file -> float32 [time] (soundfile's float convention), channels averaged
        (inference/interfaces.py:343-358 + dataio/preprocess.py:49-84)."""
        x, sr = read_wav(path)
        sig = torch.from_numpy(np.ascontiguousarray(x))
        if sig.shape[1] == 1:
            sig = sig[:, 0]
        return self.audio_normalizer(sig, sr).to(self.device)

    @classmethod
    def from_hparams(cls, source, hparams_file="hyperparams.yaml", pymodule_file="custom.py", overrides=None,
                     savedir=None, run_opts=None, freeze_params=True, **kwargs):
        """inference/interfaces.py:455-489 + utils/fetching / parameter_transfer for a local ``source`` directory:
        build every object of ``<source>/<hparams_file>`` (``speechbrain.*`` classes are served by
        ``speechbrain_amd.*``), run its ``pretrainer`` over ``<source>/<name>.ckpt`` and wrap ``modules``."""
        from speechbrain_amd.utils.hpyaml import load_hyperpyyaml

        source = str(source)
        path = os.path.join(source, hparams_file)
        if not os.path.isfile(path):
            raise FileNotFoundError(
                f"from_hparams: '{path}' not found. `source` must be a local model directory (hyperparams.yaml + "
                "checkpoints); HuggingFace ids / URLs are not fetched on the MI355X path.")
        if os.path.isfile(os.path.join(source, pymodule_file)):
            raise NotImplementedError(f"from_hparams: custom python modules ({pymodule_file}) are not executed")
        with open(path, encoding="utf-8") as f:
            hparams = load_hyperpyyaml(f, overrides)
        pretrainer = hparams.get("pretrainer")
        if pretrainer is not None:
            pretrainer.set_collect_in(savedir)
            pretrainer.collect_files(default_source=source)
            pretrainer.load_collected()
        return cls(hparams["modules"], hparams, run_opts=run_opts, freeze_params=freeze_params, **kwargs)
