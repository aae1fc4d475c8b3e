"""
Global layer normalisation on MI355X.  API of reference src/modules/norm.py:11-35 (`GlobalLayerNorm`, same
`norm.weight` / `norm.bias` state_dict keys, eps=1e-12), arithmetic in libsepkernels (sep_gln_*).

Inside ConvTasNet the norm never runs as its own kernel (it is folded into the neighbouring GEMM / depthwise
kernels, see sepkernels/net.py); this module is the stand-alone form for other callers.
`CumulativeLayerNorm1d` (causal cLN, reference :42-101) is outside the hot path (causal=0 in every BASELINE
config) and is declared only so that factories and checkpoints referring to it import cleanly.
"""
import torch
import torch.nn as nn

import sepkernels

EPS = 1e-12


class _GlobalLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        K = sepkernels.backend()
        B, C = x.shape[0], x.shape[1]
        T = x.numel() // (B * C)
        ldt = (T + 3) // 4 * 4
        x3 = x.contiguous().view(B, C, T)
        if ldt != T:
            xp = torch.empty(B, C, ldt, device=x.device, dtype=x.dtype)
            K.repack(x3, T, xp, ldt, B * C, T)
        else:
            xp = x3
        stats = torch.zeros(B, sepkernels.STATS_SLOTS, 2, device=x.device, dtype=torch.float64)
        K.gln_stats(xp, stats, B, C, T, ldt)
        yp = torch.empty_like(xp)
        K.gln_apply(xp, stats, gamma, beta, yp, B, C, T, ldt, C * T, eps)
        if ldt != T:
            y = torch.empty(B, C, T, device=x.device, dtype=x.dtype)
            K.repack(yp, ldt, y, T, B * C, T)
        else:
            y = yp
        ctx.save_for_backward(xp, stats, gamma)
        ctx.geom = (B, C, T, ldt, eps, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        K = sepkernels.backend()
        xp, stats, gamma = ctx.saved_tensors
        B, C, T, ldt, eps, shape = ctx.geom
        dy3 = dy.contiguous().view(B, C, T)
        if ldt != T:
            dyp = torch.empty_like(xp)
            K.repack(dy3, T, dyp, ldt, B * C, T)
        else:
            dyp = dy3
        ntile = (ldt + 1023) // 1024
        f32 = dict(device=xp.device, dtype=xp.dtype)
        rp = torch.empty(B, C, ntile, 2, **f32)
        K.gln_bwd_rowsums(dyp, xp, rp, B, C, T, ldt)
        bsum, pbeta, pgamma = torch.empty(B, 2, **f32), torch.empty(B, C, **f32), torch.empty(B, C, **f32)
        K.gln_bwd_finalize(rp, ntile, 2, stats, gamma, C * T, eps, bsum, pbeta, pgamma, None, B, C)
        dgamma, dbeta = torch.empty(C, **f32), torch.empty(C, **f32)
        K.reduce_slabs([(pgamma, 0, dgamma, C, B, C, 0, 1.0), (pbeta, 0, dbeta, C, B, C, 0, 1.0)])
        dxp = torch.empty_like(xp)
        K.gln_bwd_apply(dyp, xp, stats, gamma, bsum, dxp, B, C, T, ldt, C * T, eps)
        if ldt != T:
            dx = torch.empty(B, C, T, **f32)
            K.repack(dxp, ldt, dx, T, B * C, T)
        else:
            dx = dxp
        return dx.view(shape), dgamma, dbeta, None


class GlobalLayerNorm(nn.Module):
    def __init__(self, num_features, eps=EPS):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        # parameter holder with the reference's names (norm.weight, norm.bias) and default init (1, 0)
        self.norm = nn.GroupNorm(1, num_features, eps=eps)

    def forward(self, input):
        """input (batch_size, C, *) -> same shape; statistics over (C, *) per sample."""
        return _GlobalLayerNormFn.apply(input, self.norm.weight, self.norm.bias, self.eps)

    def __repr__(self):
        return "{}({}, eps={})".format(self.__class__.__name__, self.num_features, self.eps)


class CumulativeLayerNorm1d(nn.Module):
    def __init__(self, num_features, eps=EPS):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(1, num_features, 1))
        self.beta = nn.Parameter(torch.zeros(1, num_features, 1))

    def forward(self, input):
        raise NotImplementedError("CumulativeLayerNorm1d (causal Conv-TasNet) is outside the MI355X hot path "
                                  "(SURVEY.md section 8: causal=0 in every benchmark configuration)")

    def __repr__(self):
        return "{}({}, eps={})".format(self.__class__.__name__, self.num_features, self.eps)
