"""Segment1d / OverlapAdd1d with the API of reference src/models/transform.py:6-65 on the index-map kernels of
libsepkernels (sep_segment / sep_overlap_add): no unfold/fold buffers, no permute copies."""
import torch
import torch.nn as nn

from sepkernels.functional import SegmentFn, OverlapAddFn


def _padded(x):
    """(B, C, T) -> contiguous (B, C, ldt) with ldt % 4 == 0 and zero pad frames"""
    B, C, T = x.shape
    ldt = (T + 3) // 4 * 4
    if ldt == T:
        return x.contiguous(), T, ldt
    xp = torch.zeros(B, C, ldt, device=x.device, dtype=x.dtype)
    xp[:, :, :T] = x
    return xp, T, ldt


class Segment1d(nn.Module):
    def __init__(self, chunk_size, hop_size):
        super().__init__()
        self.chunk_size, self.hop_size = chunk_size, hop_size

    def forward(self, input):
        """input (B, C, n_frames) with (n_frames - chunk_size) % hop_size == 0 -> (B, C, S, chunk_size)"""
        n_frames = input.size(2)
        assert (n_frames - self.chunk_size) % self.hop_size == 0, "pad the input first (as the reference's separators do)"
        xp, T, _ = _padded(input)
        return SegmentFn.apply(xp, T, self.chunk_size, self.hop_size)

    def extra_repr(self):
        return "chunk_size={}, hop_size={}".format(self.chunk_size, self.hop_size)


class OverlapAdd1d(nn.Module):
    def __init__(self, chunk_size, hop_size):
        super().__init__()
        self.chunk_size, self.hop_size = chunk_size, hop_size

    def forward(self, input):
        """input (B, C, S, chunk_size) -> (B, C, (S-1)*hop_size + chunk_size)"""
        S = input.size(2)
        T = (S - 1) * self.hop_size + self.chunk_size
        ldt = (T + 3) // 4 * 4
        return OverlapAddFn.apply(input, T, ldt, self.hop_size)[:, :, :T]

    def extra_repr(self):
        return "chunk_size={}, hop_size={}".format(self.chunk_size, self.hop_size)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
