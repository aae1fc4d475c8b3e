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


def _activation(owner, nonlinear):
    """Registers `nonlinear1d` on `owner` when requested (scalar PReLU is the only activation of the reference TCN)."""
    owner.nonlinear = nonlinear is not None
    if owner.nonlinear:
        if nonlinear != "prelu":
            raise ValueError("Not support {}".format(nonlinear))
        owner.nonlinear1d = nn.PReLU()


def _normalisation(owner, channels, causal, eps):
    owner.norm1d = choose_layer_norm("cLN" if causal else "gLN", channels, causal=causal, eps=eps)


class _Container(nn.Module):
    """Parameter container: the arithmetic lives in sepkernels/net.py."""

    def forward(self, input):
        raise NotImplementedError(_MSG.format(type(self).__name__))


class TimeDilatedConvNet(_Container):
    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, num_blocks=3, num_layers=10,
                 dilated=True, separable=False, causal=True, nonlinear=None, norm=True, eps=EPS):
        super().__init__()
        self.num_blocks = num_blocks
        shared = dict(hidden_channels=hidden_channels, skip_channels=skip_channels, kernel_size=kernel_size, num_layers=num_layers,
                      dilated=dilated, separable=separable, causal=causal, nonlinear=nonlinear, norm=norm, eps=eps)
        # every block but the last feeds the next one, so only the last block's last layer lacks the output head
        self.net = nn.Sequential(*[TimeDilatedConvBlock1d(num_features, dual_head=(b + 1 < num_blocks), **shared) for b in range(num_blocks)])


class TimeDilatedConvBlock1d(_Container):
    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, num_layers=10, dilated=True,
                 separable=False, causal=True, nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.num_layers = num_layers
        shared = dict(hidden_channels=hidden_channels, skip_channels=skip_channels, kernel_size=kernel_size, separable=separable,
                      causal=causal, nonlinear=nonlinear, norm=norm, eps=eps)
        layers = []
        for x in range(num_layers):
            geometry = dict(dilation=2 ** x, stride=1) if dilated else dict(dilation=1, stride=2)
            layers.append(ResidualBlock1d(num_features, dual_head=(dual_head or x + 1 < num_layers), **geometry, **shared))
        self.net = nn.Sequential(*layers)


class ResidualBlock1d(_Container):
    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, stride=2, dilation=1,
                 separable=False, causal=True, nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.separable, self.causal, self.norm, self.dual_head = separable, causal, norm, dual_head
        # (creation order = the reference's, so a seeded default initialisation draws the same numbers)
        self.bottleneck_conv1d = nn.Conv1d(num_features, hidden_channels, kernel_size=1, stride=1)
        _activation(self, nonlinear)
        if norm:
            _normalisation(self, hidden_channels, causal, eps)
        if separable:
            self.separable_conv1d = DepthwiseSeparableConv1d(hidden_channels, num_features, skip_channels=skip_channels,
                                                             kernel_size=kernel_size, stride=stride, dilation=dilation, causal=causal,
                                                             nonlinear=nonlinear, norm=norm, dual_head=dual_head, eps=eps)
            return
        taps = dict(kernel_size=kernel_size, dilation=dilation)
        if dual_head:
            self.output_conv1d = nn.Conv1d(hidden_channels, num_features, **taps)
        self.skip_conv1d = nn.Conv1d(hidden_channels, skip_channels, **taps)


class DepthwiseSeparableConv1d(_Container):
    def __init__(self, in_channels, out_channels=256, skip_channels=256, kernel_size=3, stride=2, dilation=1, causal=True,
                 nonlinear=None, norm=True, dual_head=True, eps=EPS):
        super().__init__()
        self.dual_head, self.norm, self.eps = dual_head, norm, eps
        self.depthwise_conv1d = nn.Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, dilation=dilation,
                                          groups=in_channels)
        _activation(self, nonlinear)
        if norm:
            _normalisation(self, in_channels, causal, eps)
        if dual_head:
            self.output_pointwise_conv1d = nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=1)
        self.skip_pointwise_conv1d = nn.Conv1d(in_channels, skip_channels, kernel_size=1, stride=1)
