"""speechbrain.utils.data_utils slice used by inference: padding helpers (data_utils.py:28-58,459-519)."""
import torch

from speechbrain_amd.decoders.utils import undo_padding  # noqa: F401


def batch_pad_right(tensors, value=0.0):
    """Zero right-pad 1-D waveforms to the longest; returns (batch [B,N], rel_lens [B])."""
    if not len(tensors):
        raise IndexError("Tensors list must not be empty")
    n = max(t.shape[0] for t in tensors)
    out = torch.full((len(tensors), n), value, dtype=tensors[0].dtype)
    lens = []
    for i, t in enumerate(tensors):
        out[i, : t.shape[0]] = t
        lens.append(t.shape[0] / n)
    return out, torch.tensor(lens)
