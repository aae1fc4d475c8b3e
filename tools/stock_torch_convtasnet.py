"""Measurement aid for bench.py's `hipified_baseline` leg: the SAME Conv-TasNet training step written with stock
torch.nn modules only -- nn.Conv1d / nn.GroupNorm(1, C) / nn.PReLU / nn.ConvTranspose1d, autograd, torch.optim.Adam -- so that on an
MI355X it runs on MIOpen / rocBLAS / ATen's elementwise kernels, i.e. what the reference's module code does on PyTorch-ROCm without any
of this library's kernels.  It separates "what the device gives" from "what the hand-written path gives".  Nothing in the product
imports this file, and it imports nothing from oracle/ or from the product.

Structure follows the reference's modules (src/models/conv_tasnet.py:116-171,359-378; src/models/tdcn.py:13-196; src/modules/norm.py:11-29;
src/models/filterbank.py:205-251; src/criterion/sdr.py:122-139; src/criterion/pit.py:9-44), paper-best family only (trainable bases, gLN,
PReLU, depthwise-separable dilated TCN, sigmoid mask)."""
import itertools
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

EPS = 1e-12


class _Layer(nn.Module):
    def __init__(self, Bn, H, Sc, P, dilation, dual):
        super().__init__()
        self.conv1 = nn.Conv1d(Bn, H, 1)
        self.act1, self.norm1 = nn.PReLU(), nn.GroupNorm(1, H, eps=EPS)
        self.pad = (P - 1) * dilation
        self.dw = nn.Conv1d(H, H, P, dilation=dilation, groups=H)
        self.act2, self.norm2 = nn.PReLU(), nn.GroupNorm(1, H, eps=EPS)
        self.out = nn.Conv1d(H, Bn, 1) if dual else None
        self.skip = nn.Conv1d(H, Sc, 1)

    def forward(self, x):
        h = self.norm1(self.act1(self.conv1(x)))
        h = self.dw(F.pad(h, (self.pad // 2, self.pad - self.pad // 2)))
        h = self.norm2(self.act2(h))
        return (self.out(h) + x if self.out is not None else None), self.skip(h)


class StockConvTasNet(nn.Module):
    def __init__(self, n_basis, kernel_size, stride, sep_hidden_channels, sep_bottleneck_channels, sep_skip_channels, sep_kernel_size,
                 sep_num_blocks, sep_num_layers, n_sources, enc_nonlinear=None, **_):
        super().__init__()
        N, L, S = n_basis, kernel_size, stride
        self.L, self.S, self.n_src, self.N, self.relu = L, S, n_sources, N, enc_nonlinear == "relu"
        self.encoder = nn.Conv1d(1, N, L, stride=S, bias=False)
        self.norm0 = nn.GroupNorm(1, N, eps=EPS)
        self.bottleneck = nn.Conv1d(N, sep_bottleneck_channels, 1)
        R, X = sep_num_blocks, sep_num_layers
        self.layers = nn.ModuleList([_Layer(sep_bottleneck_channels, sep_hidden_channels, sep_skip_channels, sep_kernel_size, 2 ** x,
                                            not (r == R - 1 and x == X - 1)) for r in range(R) for x in range(X)])
        self.act = nn.PReLU()
        self.mask = nn.Conv1d(sep_skip_channels, n_sources * N, 1)
        self.decoder = nn.ConvTranspose1d(N, 1, L, stride=S, bias=False)

    def forward(self, x):
        B, _, T = x.shape
        padding = (self.S - (T - self.L) % self.S) % self.S
        pl, pr = padding // 2, padding - padding // 2
        w = self.encoder(F.pad(x, (pl, pr)))
        if self.relu:
            w = F.relu(w)
        h = self.bottleneck(self.norm0(w))
        skip = 0
        for layer in self.layers:
            h, s = layer(h)
            skip = skip + s
        m = torch.sigmoid(self.mask(self.act(skip))).view(B, self.n_src, self.N, -1)
        y = self.decoder((w.unsqueeze(1) * m).view(B * self.n_src, self.N, -1))
        return F.pad(y.view(B, self.n_src, -1), (-pl, -pr))


def neg_sisdr_pit(est, src, eps=EPS):
    n = est.shape[1]
    losses = []
    for pat in itertools.permutations(range(n)):
        t = src[:, list(pat)]
        alpha = (est * t).sum(-1, keepdim=True) / ((t ** 2).sum(-1, keepdim=True) + eps)
        v = (((alpha * t) ** 2).sum(-1) + eps) / (((alpha * t - est) ** 2).sum(-1) + eps)
        losses.append((-10 * torch.log10(v)).mean(1))
    return torch.stack(losses, 1).min(1)[0].mean(0)


def time_train_step(cfg, mixture, sources, steps=5, warmup=2, lr=1e-3, max_norm=5.0):
    """seconds per step of forward + PIT(NegSI-SDR) + backward + clip + Adam on mixture's device"""
    torch.manual_seed(111)
    model = StockConvTasNet(**cfg).to(mixture.device)
    opt = torch.optim.Adam(model.parameters(), lr=lr)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = neg_sisdr_pit(model(mixture), sources)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        opt.step()
    for _ in range(warmup):
        step()
    if mixture.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if mixture.is_cuda:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, sum(p.numel() for p in model.parameters())
