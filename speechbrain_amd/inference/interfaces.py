"""speechbrain.inference.interfaces mirror: the slice of ``Pretrained`` that EncoderDecoderASR needs
(inference/interfaces.py:216-489).  Construction from modules=/hparams= dicts and ``from_hparams`` on a LOCAL
model directory (hyperparams.yaml + *.ckpt, the layout of the reference's HuggingFace model cards) are
supported; fetching from HuggingFace / URLs is not (no network on the box)."""
import os
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
