"""
Waveform-distance criteria on MI355X.  API of reference src/criterion/distance.py: `L1Loss` (:7-45), `L2Loss` (:48-91),
`MeanAbsoluteError` (:207-245), `MeanSquaredError` (:247-285) with `dim`, `reduction`,
`forward(input, target, batch_mean=True)` and `.maximize`; these are the `--criterion mae|mse` choices of the
music recipe (egs/musdb18/conv-tasnet/local/train.py:117-126).

The O(T) work is two kernels of libsepkernels: sep_rowdiff_sums (sum |x-t|, sum (x-t)^2, sum t^2 per row, fp64
across the workgroup) and sep_rowdiff_bwd (dx = c_abs sign(x-t) + c_sq (x-t)); what is left on the host side is a
few scalars per row, combined exactly as the reference formula states.
"""
import math

import torch
import torch.nn as nn

import sepkernels

EPS = 1e-12

_MAX_ROWS = 65535    # grid.y limit of sep_rowdiff_bwd


class _RowDiffFn(torch.autograd.Function):
    """x, t (rows, T) -> (rows,) value of `kind` over the last axis.

    kind        value                               c_abs       c_sq
    abs_sum     sum |x-t|                           1           -
    abs_mean    sum |x-t| / T                       1/T         -
    sq_mean     sum (x-t)^2 / T                     -           2/T
    l2          sqrt(sum (x-t)^2)                   -           1/value
    sdr         10 log10((sum t^2+eps)/(sum (x-t)^2+eps))   -   -20/ln10 / (sum (x-t)^2 + eps)
    """

    @staticmethod
    def forward(ctx, x, t, kind, eps):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("gradient w.r.t. the target of a distance criterion is not implemented")
        K = sepkernels.backend()
        rows, T = x.shape
        sums = torch.empty(rows, 3, device=x.device, dtype=torch.float64)
        K.rowdiff_sums(x, t, sums, rows, T)
        s_abs, s_sq, s_tt = sums.unbind(dim=1)
        c_abs = c_sq = None
        if kind == "abs_sum":
            val, c_abs = s_abs, torch.ones_like(s_abs)
        elif kind == "abs_mean":
            val, c_abs = s_abs / T, torch.full_like(s_abs, 1.0 / T)
        elif kind == "sq_mean":
            val, c_sq = s_sq / T, torch.full_like(s_sq, 2.0 / T)
        elif kind == "l2":
            val = torch.sqrt(s_sq)
            c_sq = 1.0 / val
        elif kind == "sdr":
            val = 10.0 * torch.log10((s_tt + eps) / (s_sq + eps))
            c_sq = (-20.0 / math.log(10.0)) / (s_sq + eps)
        else:
            raise ValueError("unknown row distance '{}'".format(kind))
        ctx.save_for_backward(x, t, c_abs if c_abs is not None else c_sq)
        ctx.is_abs = c_abs is not None
        return val.to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        K = sepkernels.backend()
        x, t, coef = ctx.saved_tensors
        rows, T = x.shape
        coef = (coef * gout.double()).to(torch.float32).contiguous()
        dx = torch.empty_like(x)
        for r0 in range(0, rows, _MAX_ROWS):
            r1 = min(rows, r0 + _MAX_ROWS)
            c = coef[r0:r1]
            K.rowdiff_bwd(x[r0:r1], t[r0:r1], c if ctx.is_abs else None, None if ctx.is_abs else c, dx[r0:r1], r1 - r0, T)
        return dx, None, None, None


def row_distance(input, target, dim, kind, eps=EPS):
    """Reduce `kind` of (input, target) over `dim` (int or tuple of ints); returns the remaining axes."""
    dims = (dim,) if isinstance(dim, int) else tuple(dim)
    dims = tuple(sorted(d % input.dim() for d in dims))
    if input.dtype != torch.float32 and sepkernels.backend().name == "hip":
        input = input.float()
    target = target.to(input.dtype).expand_as(input)
    last = tuple(range(input.dim() - len(dims), input.dim()))
    if dims != last:
        input, target = input.movedim(dims, last), target.movedim(dims, last)
    lead = input.shape[:input.dim() - len(dims)]
    T = 1
    for d in last:
        T *= input.shape[d]
    rows = input.numel() // T
    out = _RowDiffFn.apply(input.reshape(rows, T).contiguous(), target.reshape(rows, T).contiguous(), kind, eps)
    return out.view(lead)


class _RowCriterion(nn.Module):
    _kind = None
    _reductions = ("mean", "sum", None)

    def __init__(self, dim=1, reduction=None):
        super().__init__()
        if reduction not in self._reductions:
            raise ValueError("Invalid reduction type")
        self.dim = dim
        self.reduction = reduction

    def forward(self, input, target, batch_mean=True):
        """
        Args:
            input, target: (batch_size, *)
        Returns:
            loss: () if batch_mean else (batch_size,) (or (batch_size, *) when reduction is None)
        """
        loss = row_distance(input, target, self.dim, self._kind)
        if self.reduction and loss.dim() > 1:
            rest = tuple(range(1, loss.dim()))
            loss = loss.mean(dim=rest) if self.reduction == "mean" else loss.sum(dim=rest)
        if batch_mean:
            loss = loss.mean(dim=0)
        return loss

    @property
    def maximize(self):
        return False


class MeanAbsoluteError(_RowCriterion):
    _kind = "abs_mean"


class MeanSquaredError(_RowCriterion):
    _kind = "sq_mean"


class L1Loss(_RowCriterion):
    _kind = "abs_sum"
    _reductions = ("mean", "sum")

    def __init__(self, dim=1, reduction="mean"):
        super().__init__(dim=dim, reduction=reduction)


class L2Loss(_RowCriterion):
    _kind = "l2"
    _reductions = ("mean", "sum")

    def __init__(self, dim=1, reduction="mean"):
        super().__init__(dim=dim, reduction=reduction)


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
