"""
`DepthwiseSeparableConv1d` of reference src/modules/conv.py:13-29 (depthwise -> pointwise, no norm/activation).
Conv-TasNet does NOT use this class (its separable block is models/tdcn.py's class of the same name); it is the
`src/conv.py` the north-star text mentions.  Parameter container with the reference's names; the depthwise half is a generic
streaming kernel (sep_depthwise_*), the pointwise half runs on the MFMA GEMM of libsepkernels.
"""
import torch.nn as nn

from sepkernels.functional import DepthwiseConv1dFn, PointwiseConv1dFn


class DepthwiseSeparableConv1d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=None, padding=0, dilation=1, bias=True):
        super().__init__()
        if stride is None:
            stride = kernel_size
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.depthwise_conv1d = nn.Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                          dilation=dilation, groups=in_channels, bias=bias)
        self.pointwise_conv1d = nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1, bias=bias)

    def forward(self, input):
        """input (B, in_channels, T) -> (B, out_channels, T')"""
        dw, pw = self.depthwise_conv1d, self.pointwise_conv1d
        x = DepthwiseConv1dFn.apply(input, dw.weight, dw.bias, dw.stride[0], dw.padding[0], dw.dilation[0])
        return PointwiseConv1dFn.apply(x, pw.weight, pw.bias)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
