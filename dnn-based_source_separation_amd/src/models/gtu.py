"""Gated tanh unit of the DPTNet / GALRNet / SepFormer mask heads (reference src/models/gtu.py:10-44): two convolutions of the
same input, tanh of one times sigmoid of the other.  With kernel size 1 both are ONE product over the stacked weights on the MFMA
GEMM of libsepkernels; any other geometry is the torch convolution pair."""
import torch
import torch.nn as nn

from sepkernels.functional import PointwiseConv1dFn, takes as _takes


class GTU1d(nn.Module):
    """Gated tanh unit, tanh(map(x)) * sigmoid(map_gate(x)); parameter names of reference src/models/gtu.py:10-44."""

    def __init__(self, in_channels, out_channels=None, kernel_size=1, stride=1, padding=0, dilation=1):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        kw = dict(kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation)
        self.map = nn.Conv1d(in_channels, out_channels, **kw)
        self.map_gate = nn.Conv1d(in_channels, out_channels, **kw)

    def stacked(self):
        """([map; map_gate] weight, bias): both halves as one 1x1 convolution"""
        return torch.cat([self.map.weight, self.map_gate.weight], 0), torch.cat([self.map.bias, self.map_gate.bias], 0)

    def forward(self, input):
        """input (B, in_channels, T) -> (B, out_channels, T')"""
        pointwise = self.map.kernel_size == (1,) and self.map.stride == (1,) and self.map.padding == (0,)
        if pointwise and _takes(input) and not (self.in_channels % 16 or self.out_channels % 16):
            W, b = self.stacked()
            ab = PointwiseConv1dFn.apply(input, W, b)
            return torch.tanh(ab[:, :self.out_channels]) * torch.sigmoid(ab[:, self.out_channels:])
        return torch.tanh(self.map(input)) * torch.sigmoid(self.map_gate(input))


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
