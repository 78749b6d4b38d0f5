"""Helpers the DUET agent imports from `models.ops` (VLN-DUET/map_nav_src/models/ops.py:25-68)."""
import torch


def extend_neg_masks(masks, dtype=None):
    """(N, L) 0/1 mask -> additive (N, 1, 1, L) with (1 - m) * -10000."""
    return ((1.0 - masks.unsqueeze(1).unsqueeze(2).to(dtype or torch.float)) * -10000.0)


def gen_seq_masks(seq_lens, max_len=None):
    max_len = int(max(seq_lens)) if max_len is None else max_len
    return torch.arange(max_len, device=seq_lens.device)[None, :] < seq_lens[:, None]


def pad_tensors_wgrad(tensors, lens=None):
    """B x [T_i, ...] -> [B, max T, ...], zero padded, differentiable (one pad per tensor instead of cat+zeros)."""
    lens = [t.size(0) for t in tensors] if lens is None else lens
    mx = max(lens)
    out = []
    for t, n in zip(tensors, lens):
        if n < mx:
            pad = [0, 0] * (t.dim() - 1) + [0, mx - n]
            t = torch.nn.functional.pad(t, pad)
        out.append(t)
    return torch.stack(out, 0)
