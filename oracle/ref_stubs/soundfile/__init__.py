"""Stand-in for the `soundfile` package (oracle/make_golden.py only): 16-bit PCM wav files are read with
the stdlib and scaled like libsndfile does for float reads (int16 / 32768)."""
import wave

import numpy as np

_ffi_types = {"float32": "float", "float64": "double", "int16": "short", "int32": "int"}


def read(path, start=0, frames=-1, dtype="float32", always_2d=False, **_):
    with wave.open(str(path), "rb") as f:
        assert f.getsampwidth() == 2, "stub: 16-bit PCM only"
        sr, ch, n = f.getframerate(), f.getnchannels(), f.getnframes()
        pcm = np.frombuffer(f.readframes(n), dtype="<i2").reshape(-1, ch)
    pcm = pcm[start:] if frames < 0 else pcm[start:start + frames]
    x = pcm.astype(np.float64) / 32768.0 if dtype.startswith("float") else pcm
    x = x.astype(dtype)
    if ch == 1 and not always_2d:
        x = x[:, 0]
    return x, sr
