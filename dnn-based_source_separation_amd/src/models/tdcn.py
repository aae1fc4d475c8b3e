"""
Time-dilated convolutional network (TCN) of Conv-TasNet: module tree, parameter names, shapes and default
initialisation of reference src/models/tdcn.py:13-196, so that reference checkpoints load unchanged.

On MI355X the whole stack of the north-star configuration family (non-causal, separable, dilated, gLN, PReLU, P = 3) runs
as the fused kernel sequence of sepkernels/net.py (three kernels per layer forward, with gLN / PReLU / residual / skip
folded into the MFMA GEMMs and the LDS-staged depthwise kernel), driven by models.conv_tasnet.ConvTasNet, and these modules
are then only its parameter containers.  Their own `forward` methods are the layer-by-layer composition (torch convolutions
on the device of the input, this repository's norm modules): what a stand-alone call runs, and what ConvTasNet falls back to
for the configurations the fused sequence does not cover (causal / cLN, non-separable, non-dilated, no norm, P != 3) --
SURVEY.md section 8b.
"""
import torch.nn as nn
import torch.nn.functional as F

from utils.tasnet import choose_layer_norm

EPS = 1e-12


def _activation(owner, nonlinear):
    """Registers `nonlinear1d` on `owner` when requested (scalar PReLU is the only activation of the reference TCN)."""
    owner.nonlinear = nonlinear is not None
    if owner.nonlinear:
        if nonlinear != "prelu":
            raise ValueError("Not support {}".format(nonlinear))
        owner.nonlinear1d = nn.PReLU()


def _normalisation(owner, channels, causal, eps):
    owner.norm1d = choose_layer_norm("cLN" if causal else "gLN", channels, causal=causal, eps=eps)


def _act_norm(owner, x):
    """nonlinear1d then norm1d, each only if the layer has it (reference tdcn.py:113-116, 182-186)."""
    if owner.nonlinear:
        x = owner.nonlinear1d(x)
    if owner.norm:
        x = owner.norm1d(x)
    return x


class TimeDilatedConvNet(nn.Module):
    def __init__(self, num_features, hidden_channels=256, skip_channels=256, kernel_size=3, num_blocks=3, num_layers=10,
                 dilated=True, separable=False, causal=True, nonlinear=None, norm=True, eps=EPS):
        super().__init__()
        self.num_blocks = num_blocks
        shared = dict(hidden_channels=hidden_channels, skip_channels=skip_channels, kernel_size=kernel_size, num_layers=num_layers,
                      dilated=dilated, separable=separable, causal=causal, nonlinear=nonlinear, norm=norm, eps=eps)
        # every block but the last feeds the next one, so only the last block's last layer lacks the output head
        self.net = nn.Sequential(*[TimeDilatedConvBlock1d(num_features, dual_head=(b + 1 < num_blocks), **shared) for b in range(num_blocks)])

    def forward(self, input):
        """input (batch_size, num_features, T) -> sum of every layer's skip output (batch_size, skip_channels, T)  (tdcn.py:29-41)"""
        x, total = input, 0
        for block in self.net:
            x, skip = block(x)
            total = total + skip
        return total


class TimeDilatedConvBlock1d(nn.Module):
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

    def forward(self, input):
        """-> (output of the last layer, or None if it has no output head ; sum of the layers' skip outputs)  (tdcn.py:65-75)"""
        x, total = input, 0
        for layer in self.net:
            x, skip = layer(x)
            total = total + skip
        return x, total


class ResidualBlock1d(nn.Module):
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

    def forward(self, input):
        """input (batch_size, num_features, T) -> (input + output head, or None ; skip head), both with T frames  (tdcn.py:107-147).
        The zero padding goes in AFTER the norm: all of it on the left when causal, split (left gets the smaller half) otherwise."""
        T = input.size(-1)
        x = _act_norm(self, self.bottleneck_conv1d(input))
        pad = (T - 1) * self.stride - T + (self.kernel_size - 1) * self.dilation + 1
        left = pad if self.causal else pad // 2
        x = F.pad(x, (left, pad - left))
        if self.separable:
            out, skip = self.separable_conv1d(x)
        else:
            out = self.output_conv1d(x) if self.dual_head else None
            skip = self.skip_conv1d(x)
        return (None if out is None else out + input), skip


class DepthwiseSeparableConv1d(nn.Module):
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

    def forward(self, input):
        """already padded input (batch_size, C, T_padded) -> (output head or None, skip head)  (tdcn.py:177-196)"""
        x = _act_norm(self, self.depthwise_conv1d(input))
        return (self.output_pointwise_conv1d(x) if self.dual_head else None), self.skip_pointwise_conv1d(x)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
