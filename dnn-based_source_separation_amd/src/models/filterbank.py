"""
Learned encoder / decoder of Conv-TasNet.  API and state_dict keys of reference src/models/filterbank.py:205-251
(`Encoder.conv1d.weight (N, in_channels, L)`, `Decoder.conv_transpose1d.weight (N, out_channels, L)`, no bias,
default nn.Conv1d / nn.ConvTranspose1d initialisation).  Inside ConvTasNet both run as part of the fused
network (sep_encoder_fwd / sep_decoder_fwd); stand-alone calls are forward-only.
"""
import torch
import torch.nn as nn

import sepkernels


def _round_up(a, b):
    return (a + b - 1) // b * b


class Encoder(nn.Module):
    def __init__(self, in_channels, n_basis, kernel_size=16, stride=8, nonlinear=None):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self.conv1d = nn.Conv1d(in_channels, n_basis, kernel_size=kernel_size, stride=stride, bias=False)   # parameter holder
        if nonlinear is not None and nonlinear != "":
            if nonlinear != "relu":
                raise NotImplementedError("Not support {}".format(nonlinear))
            self.nonlinear1d = nn.ReLU()
            self.nonlinear = True
        else:
            self.nonlinear = False

    def forward(self, input):
        """input (B, in_channels, T) -> (B, n_basis, (T-L)//S+1).  Stand-alone use is inference only."""
        if torch.is_grad_enabled() and (input.requires_grad or self.conv1d.weight.requires_grad):
            raise NotImplementedError("stand-alone Encoder is forward-only (wrap in torch.no_grad()); training goes through ConvTasNet")
        K = sepkernels.backend()
        x = input.contiguous()
        B, Cin, T = x.shape
        N, L, S = self.conv1d.out_channels, self.kernel_size, self.stride
        F = (T - L) // S + 1
        ldt = _round_up(F, 128)
        w = torch.empty(B, N, ldt, device=x.device, dtype=x.dtype)
        stats = torch.zeros(B, sepkernels.STATS_SLOTS, 2, device=x.device, dtype=torch.float64)
        K.encoder_fwd(x, self.conv1d.weight, w, stats, B, Cin, T, N, L, S, F, ldt, 0, self.nonlinear)
        out = torch.empty(B, N, F, device=x.device, dtype=x.dtype)
        K.repack(w, ldt, out, F, B * N, F)
        return out

    def get_basis(self):
        return self.conv1d.weight


class Decoder(nn.Module):
    def __init__(self, n_basis, out_channels, kernel_size=16, stride=8):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self.conv_transpose1d = nn.ConvTranspose1d(n_basis, out_channels, kernel_size=kernel_size, stride=stride, bias=False)

    def forward(self, input):
        """input (B', n_basis, F) -> (B', out_channels, (F-1)*S+L).  Stand-alone use is inference only."""
        if torch.is_grad_enabled() and (input.requires_grad or self.conv_transpose1d.weight.requires_grad):
            raise NotImplementedError("stand-alone Decoder is forward-only (wrap in torch.no_grad()); training goes through ConvTasNet")
        K = sepkernels.backend()
        x = input.contiguous()
        Bp, N, F = x.shape
        L, S = self.kernel_size, self.stride
        Cout = self.conv_transpose1d.out_channels
        ldt = _round_up(F, 128)
        wp = torch.empty(Bp, N, ldt, device=x.device, dtype=x.dtype)
        K.repack(x, F, wp, ldt, Bp * N, F)
        ones = torch.ones(Bp, N, ldt, device=x.device, dtype=x.dtype)   # mask == 1: plain synthesis + overlap-add
        Tout = (F - 1) * S + L
        out = torch.empty(Bp, 1, Cout, Tout, device=x.device, dtype=x.dtype)
        K.decoder_fwd(wp, ones, self.conv_transpose1d.weight, out, None, Bp, 1, N, Cout, L, S, F, ldt, Tout, 0)
        return out.view(Bp, Cout, Tout)

    def get_basis(self):
        return self.conv_transpose1d.weight
