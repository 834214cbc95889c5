"""speechbrain.inference.interfaces mirror: the slice of ``Pretrained`` that EncoderDecoderASR needs
(inference/interfaces.py:216-489).  Construction from modules=/hparams= dicts is supported; the
HyperPyYAML / HuggingFace fetching front-end (from_hparams) is outside this round's scope."""
import types
import wave

import numpy as np
import torch


class AudioNormalizer:
    """dataio/preprocess.py:8-84 for the already-16-kHz case: channel mean; resampling is identity when
    the rates agree (augment/time_domain.py:576-577) and unsupported otherwise."""

    def __init__(self, sample_rate=16000, mix="avg-to-mono"):
        self.sample_rate, self.mix = sample_rate, mix

    def __call__(self, audio, sample_rate):
        if sample_rate != self.sample_rate:
            raise NotImplementedError(f"resampling {sample_rate} -> {self.sample_rate} Hz is not implemented")
        if audio.dim() == 2:
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
        """PCM16 wav -> float32 [time] in [-1,1) (soundfile's float convention), mono."""
        with wave.open(str(path), "rb") as f:
            if f.getsampwidth() != 2:
                raise NotImplementedError("only 16-bit PCM wav files are read by the built-in loader")
            sr, ch = f.getframerate(), f.getnchannels()
            pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")
        sig = torch.from_numpy(pcm.astype(np.float32) / 32768.0)
        if ch > 1:
            sig = sig.view(-1, ch)
        return self.audio_normalizer(sig, sr).to(self.device)

    @classmethod
    def from_hparams(cls, *args, **kwargs):
        raise NotImplementedError(
            "from_hparams needs HyperPyYAML + checkpoint fetching, which is outside this round's scope; build the "
            "modules (speechbrain_amd.inference.builders) and call cls(modules=..., hparams=...)")
