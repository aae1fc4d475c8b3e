"""
Global layer normalisation on MI355X.  API of reference src/modules/norm.py:11-35 (`GlobalLayerNorm`, same
`norm.weight` / `norm.bias` state_dict keys, eps=1e-12), arithmetic in libsepkernels (sep_gln_*).

Inside ConvTasNet the norm never runs as its own kernel (it is folded into the neighbouring GEMM / depthwise
kernels, see sepkernels/net.py); this module is the stand-alone form for other callers.
`CumulativeLayerNorm1d` (causal cLN, reference :42-101; causal=0 in every BASELINE config) runs on sep_cln_fwd / sep_cln_bwd
(csrc/cln.hip: column sums, fp64 prefix sums, one apply pass each way) for fp32 tensors the backend takes, and as the same
arithmetic composed from torch operations otherwise (float64 checks, CPU tensors in the fallback path's CPU tests).
"""
import torch
import torch.nn as nn

import sepkernels
from sepkernels.functional import cln_workspace

EPS = 1e-12
GRID_ROWS = 65535      # rows one launch of the gLN / cLN kernels takes (grid.y)


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
        if not input.is_cuda and sepkernels.backend().name == "hip":
            # CPU tensors (the reference's `--use_cuda 0` evaluation / demo mode, egs/wsj0-mix/conv-tasnet/local/test.py:25,41-43): the
            # reference's own arithmetic, nn.GroupNorm(1, C) of src/modules/norm.py:18,27.  Never taken for a tensor on the GPU.
            return self.norm(input)
        per_launch = max(1, GRID_ROWS // self.num_features)       # one grid row per (sample, channel): the dual-path models of
        if input.shape[0] <= per_launch:                          # models/{dptnet,galr,sepformer}.py normalise thousands of short samples
            return _GlobalLayerNormFn.apply(input, self.norm.weight, self.norm.bias, self.eps)
        return torch.cat([_GlobalLayerNormFn.apply(part, self.norm.weight, self.norm.bias, self.eps)
                          for part in input.split(per_launch, dim=0)], dim=0)

    def __repr__(self):
        return "{}({}, eps={})".format(self.__class__.__name__, self.num_features, self.eps)


class _CumulativeLayerNormFn(torch.autograd.Function):
    """cLN on libsepkernels (sep_cln_fwd / sep_cln_bwd, csrc/cln.hip): column sums, fp64 prefix sums, one apply pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        K = sepkernels.backend()
        B, C = x.shape[0], x.shape[1]
        T = x.numel() // (B * C)
        ldt = (T + 3) // 4 * 4
        x3 = x.contiguous().view(B, C, T)
        f32 = dict(device=x.device, dtype=torch.float32)
        if ldt != T:
            xp = torch.empty(B, C, ldt, **f32)
            K.repack(x3, T, xp, ldt, B * C, T)
        else:
            xp = x3
        yp = torch.empty_like(xp)
        mean, rstd = torch.empty(B, ldt, **f32), torch.empty(B, ldt, **f32)      # rows of ldt (ABI 20)
        ws = cln_workspace(K, B, C, T, ldt, x.device)
        g1, b1 = gamma.reshape(C).contiguous(), beta.reshape(C).contiguous()
        K.cln_fwd(xp, g1, b1, yp, mean, rstd, ws, B, C, T, ldt, eps)
        if ldt != T:
            y = torch.empty(B, C, T, **f32)
            K.repack(yp, ldt, y, T, B * C, T)
        else:
            y = yp
        ctx.save_for_backward(xp, g1, mean, rstd)
        ctx.geom = (B, C, T, ldt, eps, x.shape, gamma.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        K = sepkernels.backend()
        xp, g1, mean, rstd = ctx.saved_tensors
        B, C, T, ldt, eps, shape, pshape = ctx.geom
        f32 = dict(device=xp.device, dtype=torch.float32)
        dy3 = dy.contiguous().view(B, C, T)
        if ldt != T:
            dyp = torch.empty_like(xp)
            K.repack(dy3, T, dyp, ldt, B * C, T)
        else:
            dyp = dy3
        dxp = torch.empty_like(xp)
        pg, pb = torch.empty(B, C, **f32), torch.empty(B, C, **f32)
        ws = cln_workspace(K, B, C, T, ldt, xp.device)
        K.cln_bwd(dyp, xp, g1, mean, rstd, dxp, pg, pb, ws, B, C, T, ldt, eps)
        dgamma, dbeta = torch.empty(C, **f32), torch.empty(C, **f32)
        K.reduce_slabs([(pg, 0, dgamma, C, B, C, 0, 1.0), (pb, 0, dbeta, C, B, C, 0, 1.0)])
        if ldt != T:
            dx = torch.empty(B, C, T, **f32)
            K.repack(dxp, ldt, dx, T, B * C, T)
        else:
            dx = dxp
        return dx.view(shape), dgamma.view(pshape), dbeta.view(pshape), None


class CumulativeLayerNorm1d(nn.Module):
    def __init__(self, num_features, eps=EPS):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(1, num_features, 1))
        self.beta = nn.Parameter(torch.zeros(1, num_features, 1))

    def forward(self, input):
        """input (batch_size, C, T) or (batch_size, C, S, chunk_size) -> same shape.  Frame t is normalised with the mean and
        the (biased) variance of everything up to and including frame t: C * (t + 1) values."""
        if input.dim() not in (3, 4):
            raise ValueError("Only support 3D or 4D input, but given {}D".format(input.dim()))
        if input.dtype == torch.float32 and (input.is_cuda or sepkernels.backend().name != "hip"):
            if input.shape[0] <= GRID_ROWS:
                return _CumulativeLayerNormFn.apply(input, self.gamma, self.beta, self.eps)
            return torch.cat([_CumulativeLayerNormFn.apply(part, self.gamma, self.beta, self.eps) for part in input.split(GRID_ROWS, dim=0)], dim=0)
        return self._compose(input)

    def _compose(self, input):
        """the same arithmetic as a torch composition: float64 inputs (the CPU reference checks) and tensors the backend
        does not take (CPU tensors on the HIP build, used by the fallback path's tests)"""
        shape = input.shape
        x = input.reshape(shape[0], shape[1], -1)
        C, T = x.shape[1], x.shape[2]
        count = torch.arange(1, T + 1, device=x.device, dtype=x.dtype) * C                 # values seen after frame t
        mean = x.sum(dim=1).cumsum(dim=1) / count                                           # (batch_size, T)
        var = (x * x).sum(dim=1).cumsum(dim=1) / count - mean * mean
        y = (x - mean.unsqueeze(1)) / (var.sqrt().unsqueeze(1) + self.eps) * self.gamma + self.beta
        return y.reshape(shape)

    def __repr__(self):
        return "{}({}, eps={})".format(self.__class__.__name__, self.num_features, self.eps)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
