"""
`DepthwiseSeparableConv1d` of reference src/modules/conv.py:13-29 (depthwise -> pointwise, no norm/activation).
Conv-TasNet does NOT use this class (its separable block is models/tdcn.py's class of the same name); it is the
`src/conv.py` the north-star text mentions.  Parameter container with the reference's names; the pointwise half
runs on the MFMA GEMM of libsepkernels, the generic strided/padded depthwise half is not on the SURVEY section 8
path and is not implemented yet.
"""
import torch.nn as nn


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
        raise NotImplementedError("modules.conv.DepthwiseSeparableConv1d: generic strided depthwise is not part of the "
                                  "Conv-TasNet hot path (SURVEY.md section 8 row a14) and has no HIP kernel yet")
