"""
SDR and scale-invariant SDR on MI355X.  API of reference src/criterion/sdr.py:6-231 (`sdr`, `SDR`, `NegSDR`, `sisdr`,
`SISDR`, `NegSISDR` with `reduction`, `eps`, `forward(input, target, batch_mean=True)`, `.maximize`).

The O(T) work is two kernels of libsepkernels: sep_sisdr_dots (the three dot products per pair, fp64
accumulation) and sep_sisdr_bwd (analytic gradient applied elementwise); the value itself is formed from the
dot products exactly as the reference formula states (eps placement included).
"""
import torch
import torch.nn as nn

import sepkernels

EPS = 1e-12


_MAX_ROWS = 65535      # sep_sisdr_dots / sep_sisdr_bwd put the batch on a 16-bit grid dimension: larger batches (sisdr() flattens every
                       # leading axis into it: 4-D inputs, long evaluation lists) go through in slices


class _SISDRPairsFn(torch.autograd.Function):
    """est, tgt (B, n, T) -> sisdr (B, n, n): entry [b, i, j] = SI-SDR(est_i, tgt_j); only the diagonal is computed
    (others 0) when all_pairs is False."""

    @staticmethod
    def forward(ctx, est, tgt, all_pairs, eps):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. the SI-SDR target is not implemented")
        K = sepkernels.backend()
        est, tgt = est.contiguous(), tgt.contiguous()
        B, n, T = est.shape
        dev = est.device
        dots = torch.zeros(B, n, n, device=dev, dtype=torch.float64)
        tt = torch.zeros(B, n, device=dev, dtype=torch.float64)
        xx = torch.zeros(B, n, device=dev, dtype=torch.float64)
        out = torch.empty(B, n, n, device=dev, dtype=est.dtype)
        for b0 in range(0, B, _MAX_ROWS):
            b1 = min(B, b0 + _MAX_ROWS)
            K.sisdr_dots(est[b0:b1], tgt[b0:b1], dots[b0:b1], tt[b0:b1], xx[b0:b1], b1 - b0, n, T, all_pairs)
            K.sisdr_from_dots(dots[b0:b1], tt[b0:b1], xx[b0:b1], out[b0:b1], b1 - b0, n, all_pairs, eps)
        ctx.save_for_backward(est, tgt, dots, tt, xx)
        ctx.meta = (all_pairs, eps)
        return out

    @staticmethod
    def backward(ctx, gout):
        K = sepkernels.backend()
        est, tgt, dots, tt, xx = ctx.saved_tensors
        all_pairs, eps = ctx.meta
        B, n, T = est.shape
        d_est = torch.empty_like(est)
        gout = gout.contiguous().to(est.dtype)
        for b0 in range(0, B, _MAX_ROWS):
            b1 = min(B, b0 + _MAX_ROWS)
            K.sisdr_bwd(est[b0:b1], tgt[b0:b1], dots[b0:b1], tt[b0:b1], xx[b0:b1], gout[b0:b1], d_est[b0:b1], b1 - b0, n, T, all_pairs, eps)
        return d_est, None, None, None


def _prep(input, target):
    if input.shape != target.shape:        # the reference's formulas broadcast: its tester scores the mixture (B, 1, T) against (B, n, T) sources
        input, target = torch.broadcast_tensors(input, target)
    if input.dtype != torch.float32 and sepkernels.backend().name == "hip":
        input = input.float()
    return input, target.to(input.dtype)


def _on_host(input):
    """a tensor the caller left on the CPU while the HIP library is the backend (`--use_cuda 0` evaluation, demo notebooks): the
    criterion is then evaluated with ATen in the reference's own formula -- never for a tensor on the GPU"""
    return (not input.is_cuda) and sepkernels.backend().name == "hip"


def _sisdr_aten(input, target, eps):
    """reference src/criterion/sdr.py:135-137: projection of the estimate on the target, energy ratio in dB, eps in both sums"""
    tt = target.square().sum(-1, keepdim=True) + eps
    proj = (input * target).sum(-1, keepdim=True) / tt * target
    return 10 * torch.log10((proj.square().sum(-1) + eps) / ((proj - input).square().sum(-1) + eps))


def sisdr_pairs(input, target, eps=EPS):
    """(B, n, T) x (B, n, T) -> (B, n, n) matrix of SI-SDR(input_i, target_j)."""
    if _on_host(input):
        return _sisdr_aten(input.unsqueeze(2), target.to(input.dtype).unsqueeze(1), eps)
    input, target = _prep(input, target)
    return _SISDRPairsFn.apply(input, target, True, eps)


def sisdr(input, target, eps=EPS):
    """
    Args:
        input, target: (batch_size, T) or (batch_size, n_sources, T) or (batch_size, n_sources, n_mics, T)
    Returns:
        SI-SDR over the last axis, shape input.shape[:-1]
    """
    n_dims = input.dim()
    assert n_dims in [2, 3, 4], "Only 2D or 3D or 4D tensor is acceptable, but given {}D tensor.".format(n_dims)
    if _on_host(input):
        return _sisdr_aten(input, target.to(input.dtype), eps)
    input, target = _prep(input, target)
    lead, T = input.shape[:-1], input.shape[-1]
    rows = input.numel() // T
    out = _SISDRPairsFn.apply(input.reshape(rows, 1, T), target.reshape(rows, 1, T), False, eps)
    return out.view(lead)


def sdr(input, target, eps=EPS):
    """
    Args:
        input, target: (batch_size, T) or (batch_size, n_sources, T) or (batch_size, n_sources, n_mics, T)
    Returns:
        10 log10((|target|^2 + eps) / (|target - input|^2 + eps)) over the last axis, shape input.shape[:-1]
    """
    from criterion.distance import row_distance
    n_dims = input.dim()
    assert n_dims in [2, 3, 4], "Only 2D or 3D or 4D tensor is acceptable, but given {}D tensor.".format(n_dims)
    return row_distance(input, target, n_dims - 1, "sdr", eps=eps)


class _SISDRBase(nn.Module):
    _sign = 1.0
    _measure = staticmethod(lambda input, target, eps: sisdr(input, target, eps=eps))

    def __init__(self, reduction="mean", eps=EPS):
        super().__init__()
        if reduction not in ["mean", "sum", None]:
            raise ValueError("Invalid reduction type")
        self.reduction = reduction
        self.eps = eps

    def _clip(self, per_source):
        """hook of the clipped variants: applied to the value of every (estimate, target) pair before any reduction"""
        return per_source

    def forward(self, input, target, batch_mean=True):
        n_dims = input.dim()
        assert n_dims in [2, 3, 4], "Only 2D or 3D or 4D tensor is acceptable, but given {}D tensor.".format(n_dims)
        loss = self._clip(self._sign * self._measure(input, target, self.eps))
        if self.reduction:
            dims = {3: 1, 4: (1, 2)}.get(n_dims)
            if dims is not None:
                loss = loss.mean(dim=dims) if self.reduction == "mean" else loss.sum(dim=dims)
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss


class SISDR(_SISDRBase):
    _sign = 1.0

    @property
    def maximize(self):
        return True


class NegSISDR(_SISDRBase):
    _sign = -1.0

    @property
    def maximize(self):
        return False


class SDR(_SISDRBase):
    _sign = 1.0
    _measure = staticmethod(lambda input, target, eps: sdr(input, target, eps=eps))

    @property
    def maximize(self):
        return True


class NegSDR(_SISDRBase):
    _sign = -1.0
    _measure = staticmethod(lambda input, target, eps: sdr(input, target, eps=eps))

    @property
    def maximize(self):
        return False


class ClippedSISDR(SISDR):
    """SI-SDR with every per-source value capped at `max` dB before the reductions (reference sdr.py:233-279)"""

    def __init__(self, max=None, reduction="mean", eps=EPS):
        super().__init__(reduction=reduction, eps=eps)
        self.max = max

    def _clip(self, per_source):
        return torch.clamp(per_source, max=self.max)


class ClippedNegSISDR(NegSISDR):
    """-SI-SDR with every per-source value floored at `min` (the SepFormer recipe's criterion: min = -30 dB;
    reference sdr.py:281-327, egs/wsj0-mix/sepformer/local/train.py:125-126)"""

    def __init__(self, min=None, reduction="mean", eps=EPS):
        super().__init__(reduction=reduction, eps=eps)
        self.min = min

    def _clip(self, per_source):
        return torch.clamp(per_source, min=self.min)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
