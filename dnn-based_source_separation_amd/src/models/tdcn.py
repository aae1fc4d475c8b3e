"""
Time-dilated convolutional network (TCN) of Conv-TasNet: module tree, parameter names, shapes and default
initialisation of reference src/models/tdcn.py:13-196, so that reference checkpoints load unchanged.

The modules are parameter containers: on MI355X the whole stack runs as the fused kernel sequence of
sepkernels/net.py (three kernels per layer forward, with gLN / PReLU / residual / skip folded into the MFMA
GEMMs and the LDS-staged depthwise kernel), driven by models.conv_tasnet.ConvTasNet.  Calling a TCN sub-module
on its own is not part of the separation path and raises.
"""
import torch.nn as nn

from utils.tasnet import choose_layer_norm

EPS = 1e-12

_MSG = "{} is executed as part of the fused ConvTasNet kernel sequence (sepkernels/net.py); stand-alone forward is not implemented"


class TimeDilatedConvNet(nn.Module):
    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, num_blocks=3, num_layers=10,
                 dilated=True, separable=False, causal=True, nonlinear=None, norm=True, eps=EPS):
        super().__init__()
        self.num_blocks = num_blocks
        net = []
        for idx in range(num_blocks):
            net.append(TimeDilatedConvBlock1d(num_features, hidden_channels=hidden_channels, skip_channels=skip_channels,
                                              kernel_size=kernel_size, num_layers=num_layers, dilated=dilated, separable=separable,
                                              causal=causal, nonlinear=nonlinear, norm=norm, dual_head=(idx != num_blocks - 1), eps=eps))
        self.net = nn.Sequential(*net)

    def forward(self, input):
        raise NotImplementedError(_MSG.format("TimeDilatedConvNet"))


class TimeDilatedConvBlock1d(nn.Module):
    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, num_layers=10, dilated=True,
                 separable=False, causal=True, nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.num_layers = num_layers
        net = []
        for idx in range(num_layers):
            dilation, stride = (2 ** idx, 1) if dilated else (1, 2)
            head = dual_head or idx != num_layers - 1
            net.append(ResidualBlock1d(num_features, hidden_channels=hidden_channels, skip_channels=skip_channels,
                                       kernel_size=kernel_size, stride=stride, dilation=dilation, separable=separable,
                                       causal=causal, nonlinear=nonlinear, norm=norm, dual_head=head, eps=eps))
        self.net = nn.Sequential(*net)

    def forward(self, input):
        raise NotImplementedError(_MSG.format("TimeDilatedConvBlock1d"))


class ResidualBlock1d(nn.Module):
    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, stride=2, dilation=1,
                 separable=False, causal=True, nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.separable, self.causal = separable, causal
        self.norm = norm
        self.dual_head = dual_head
        self.bottleneck_conv1d = nn.Conv1d(num_features, hidden_channels, kernel_size=1, stride=1)
        if nonlinear is not None:
            if nonlinear != "prelu":
                raise ValueError("Not support {}".format(nonlinear))
            self.nonlinear1d = nn.PReLU()
            self.nonlinear = True
        else:
            self.nonlinear = False
        if norm:
            norm_name = "cLN" if causal else "gLN"
            self.norm1d = choose_layer_norm(norm_name, hidden_channels, causal=causal, eps=eps)
        if separable:
            self.separable_conv1d = DepthwiseSeparableConv1d(hidden_channels, num_features, skip_channels=skip_channels,
                                                             kernel_size=kernel_size, stride=stride, dilation=dilation, causal=causal,
                                                             nonlinear=nonlinear, norm=norm, dual_head=dual_head, eps=eps)
        else:
            if dual_head:
                self.output_conv1d = nn.Conv1d(hidden_channels, num_features, kernel_size=kernel_size, dilation=dilation)
            self.skip_conv1d = nn.Conv1d(hidden_channels, skip_channels, kernel_size=kernel_size, dilation=dilation)

    def forward(self, input):
        raise NotImplementedError(_MSG.format("ResidualBlock1d"))


class DepthwiseSeparableConv1d(nn.Module):
    def __init__(self, in_channels, out_channels=256, skip_channels=256, kernel_size=3, stride=2, dilation=1, causal=True,
                 nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.dual_head = dual_head
        self.norm = norm
        self.eps = eps
        self.depthwise_conv1d = nn.Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, dilation=dilation,
                                          groups=in_channels)
        if nonlinear is not None:
            if nonlinear != "prelu":
                raise ValueError("Not support {}".format(nonlinear))
            self.nonlinear1d = nn.PReLU()
            self.nonlinear = True
        else:
            self.nonlinear = False
        if norm:
            norm_name = "cLN" if causal else "gLN"
            self.norm1d = choose_layer_norm(norm_name, in_channels, causal=causal, eps=eps)
        if dual_head:
            self.output_pointwise_conv1d = nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1)
        self.skip_pointwise_conv1d = nn.Conv1d(in_channels, skip_channels, kernel_size=1, stride=1)

    def forward(self, input):
        raise NotImplementedError(_MSG.format("tdcn.DepthwiseSeparableConv1d"))
