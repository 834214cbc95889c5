"""speechbrain.decoders.utils mirror (decoders/utils.py:14-158): small tensor helpers."""
import torch


def inflate_tensor(tensor, times, dim):
    """utils.py:35-62."""
    return torch.repeat_interleave(tensor, times, dim=dim)


def mask_by_condition(tensor, cond, fill_value):
    """utils.py:65-95: keep where cond, else fill_value."""
    return torch.where(cond, tensor, torch.tensor([fill_value], device=tensor.device, dtype=tensor.dtype))


def undo_padding(batch, lengths):
    """utils/data_utils.py:28-58."""
    batch_max_len = batch.shape[1]
    out = []
    for seq, seq_length in zip(batch, lengths):
        actual = int(torch.round(seq_length * batch_max_len))
        out.append(seq.narrow(0, 0, actual).tolist())
    return out
