"""
Permutation-invariant training criteria on MI355X.  API of reference src/criterion/pit.py:9-85 (`pit`, `PIT`,
`PIT1d`, `PIT2d`) and :163-213 (`sinkpit`, `SinkPIT`): `criterion(input, target, batch_mean) -> (loss, pattern)`.

With an SI-SDR criterion (criterion.sdr.SISDR / NegSISDR) the n x n pair matrix is produced by ONE pass over the
waveforms (sep_sisdr_dots) instead of the reference's n! (PIT) or n^2 (SinkPIT) full criterion evaluations; the
permutation search (sep_pit_search) and the log-domain Sinkhorn iterations with their reverse sweep
(sep_sinkhorn_fwd/bwd) run on the n x n matrix.  Any other callable criterion takes the generic route, which
evaluates it once per permutation exactly like the reference.
"""
import itertools

import torch
import torch.nn as nn

import sepkernels
from criterion.sdr import SISDR, NegSISDR, sisdr_pairs


def _is_sisdr(criterion, input):
    """the n x n pair matrix + sep_pit_search route: SI-SDR criteria on tensors the backend takes (CPU tensors beside the HIP library --
    `--use_cuda 0` evaluation -- go the generic way below: the criterion once per permutation, like the reference)"""
    if not input.is_cuda and sepkernels.backend().name == "hip":
        return False
    return isinstance(criterion, (SISDR, NegSISDR)) and input.dim() == 3 and criterion.reduction in ("mean", "sum")


_DEVICE_PATTERNS = {}


def _on_device(patterns, device):
    """(int64, int32) device copies of a permutation table, cached: a pageable host-to-device copy per step is a
    host-side synchronisation in the middle of the step (the host would wait for the whole forward pass)."""
    if patterns.device == device:
        return patterns, patterns.to(torch.int32).contiguous()
    key = (str(device), tuple(patterns.shape))
    hit = _DEVICE_PATTERNS.get(key)
    if hit is None or not torch.equal(hit[0], patterns):
        hit = (patterns.clone(), patterns.to(device), patterns.to(device=device, dtype=torch.int32).contiguous())
        _DEVICE_PATTERNS[key] = hit
    return hit[1], hit[2]


def _fused_pit(criterion, input, target, patterns, batch_mean):
    K = sepkernels.backend()
    B, n, _ = input.shape
    maximize = bool(criterion.maximize)
    val = sisdr_pairs(input, target, eps=criterion.eps)            # (B, n, n) SI-SDR
    if not maximize:
        val = -val                                                 # NegSISDR values
    val = criterion._clip(val)                                     # clipped variants: per pair, before the mean over sources
    P = patterns.size(0)
    perms64, perms32 = _on_device(patterns, input.device)
    best_val = torch.empty(B, device=input.device, dtype=val.dtype)
    best_idx = torch.empty(B, device=input.device, dtype=torch.int64)
    K.pit_search(val.detach().contiguous(), perms32, P, n, B, maximize, criterion.reduction == "mean", best_val, best_idx)
    chosen = perms64[best_idx]                                     # (B, n)
    picked = torch.gather(val, 2, chosen.unsqueeze(2)).squeeze(2)  # val[b, s, chosen[b, s]]
    loss = picked.mean(dim=1) if criterion.reduction == "mean" else picked.sum(dim=1)
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, chosen


def pit(criterion, input, target, n_sources=None, patterns=None, batch_mean=True):
    """
    Args:
        criterion <callable>: criterion(input, target, batch_mean=False) -> (batch_size,)
        input, target (batch_size, n_sources, *)
    Returns:
        loss: () or (batch_size,) best loss per item (min, or max if criterion.maximize)
        pattern (batch_size, n_sources): chosen permutation of the targets
    """
    if patterns is None:
        if n_sources is None:
            n_sources = input.size(1)
        patterns = torch.tensor(list(itertools.permutations(range(n_sources))), dtype=torch.long)
    if input.shape != target.shape and input.dim() == target.dim():
        input, target = torch.broadcast_tensors(input, target)      # e.g. the mixture (B, 1, T) scored against (B, n, T) sources (driver.py:283)
    if _is_sisdr(criterion, input):
        return _fused_pit(criterion, input, target, patterns, batch_mean)
    # generic criterion: one evaluation per permutation
    scores = torch.stack([criterion(input, target[:, pat.to(target.device)], batch_mean=False) for pat in patterns], dim=1)
    if getattr(criterion, "maximize", False):
        loss, indices = torch.max(scores, dim=1)
    else:
        loss, indices = torch.min(scores, dim=1)
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, patterns.to(indices.device)[indices]


class PIT(nn.Module):
    def __init__(self, criterion, n_sources):
        super().__init__()
        self.criterion = criterion
        self.patterns = torch.tensor(list(itertools.permutations(range(n_sources))), dtype=torch.long)

    def forward(self, input, target, batch_mean=True):
        return pit(self.criterion, input, target, patterns=self.patterns, batch_mean=batch_mean)


class PIT1d(PIT):
    pass


class PIT2d(PIT):
    pass


class _SinkhornFn(torch.autograd.Function):
    """C (B, n, n) -> (loss (B,), P (B, n, n)); differentiates through every iteration like the reference's tape."""

    @staticmethod
    def forward(ctx, C, coldness, iteration):
        K = sepkernels.backend()
        C = C.contiguous()
        B, n, _ = C.shape
        zwork = torch.empty(B, 2 * iteration + 1, n, n, device=C.device, dtype=torch.float64)
        loss = torch.empty(B, device=C.device, dtype=C.dtype)
        P = torch.empty(B, n, n, device=C.device, dtype=C.dtype)
        K.sinkhorn_fwd(C, zwork, loss, P, B, n, float(coldness), int(iteration))
        ctx.save_for_backward(C, zwork)
        ctx.meta = (float(coldness), int(iteration))
        ctx.mark_non_differentiable(P)
        return loss, P

    @staticmethod
    def backward(ctx, dloss, _dP):
        K = sepkernels.backend()
        C, zwork = ctx.saved_tensors
        coldness, iteration = ctx.meta
        B, n, _ = C.shape
        dC = torch.empty_like(C)
        K.sinkhorn_bwd(C, zwork, dloss.contiguous().to(C.dtype), dC, B, n, coldness, iteration)
        return dC, None, None


def sinkpit(criterion, input, target, n_sources=None, coldness=1e+0, iteration=10, batch_mean=True):
    if n_sources is None:
        n_sources = input.size(1)
    maximize = bool(getattr(criterion, "maximize", False))
    if isinstance(criterion, (SISDR, NegSISDR)) and input.dim() == 3:
        # pairwise criterion on 2-D inputs has no source reduction: C_ij = -SI-SDR(input_i, target_j) for both classes
        C = -sisdr_pairs(input, target, eps=criterion.eps)
    else:
        B = input.size(0)
        xi = input.unsqueeze(2).expand(-1, -1, n_sources, *input.shape[2:]).reshape(B * n_sources * n_sources, *input.shape[2:])
        tj = target.unsqueeze(1).expand(-1, n_sources, *target.shape[1:]).reshape(B * n_sources * n_sources, *target.shape[2:])
        C = criterion(xi, tj, batch_mean=False).view(B, n_sources, n_sources)
        if maximize:
            C = -C
    loss, P = _SinkhornFn.apply(C, coldness, iteration)
    if maximize:
        loss = -loss
    if batch_mean:
        loss = loss.mean(dim=0)
    return loss, P


class SinkPIT(nn.Module):
    """Sinkhorn PIT (https://arxiv.org/abs/2010.11871), reference pit.py:195-213."""

    def __init__(self, criterion, n_sources=None, coldness=1, iteration=10):
        super().__init__()
        self.criterion = criterion
        self.n_sources = n_sources
        self.coldness = coldness
        self.iteration = iteration

    def forward(self, input, target, batch_mean=True):
        loss, permutation_matrix = sinkpit(self.criterion, input, target, n_sources=self.n_sources, coldness=self.coldness,
                                           iteration=self.iteration, batch_mean=batch_mean)
        return loss, torch.argmax(permutation_matrix, dim=2)


class ORPIT(nn.Module):
    """
    One-and-Rest permutation invariant training (reference pit.py:87-161): for every item the "one" output is matched
    against each source in turn and the "rest" output against the sum of the others,
        loss_idx = criterion(input_one, target_idx) + criterion(input_rest, sum_{j != idx} target_j) / (n_sources - 1),
    and the best candidate (min, or max if criterion.maximize) is kept.  The reference evaluates the criterion
    2 * n_sources times per item inside a python loop over the batch; here all candidates of the whole batch go through
    ONE criterion call on (n_candidates, T) tensors (with the SI-SDR criteria: one sep_sisdr_dots launch).
    """

    def __init__(self, criterion):
        super().__init__()
        self.criterion = criterion
        self.patterns = torch.tensor(list(itertools.permutations(range(2))), dtype=torch.long)

    def forward(self, input, target, batch_mean=True):
        """
        Args:
            input (batch_size, 2, *)
            target (batch_size, n_sources, *) tensor, or a PackedSequence when n_sources differs per item
        Returns:
            loss () or (batch_size,), indices (batch_size,)
        """
        assert input.size(1) == 2, "input.size() is expected (batch_size, 2, *), but given {}".format(input.size())
        if isinstance(target, torch.Tensor):
            lens = [target.size(1)] * target.size(0)
        else:
            target, lens = nn.utils.rnn.pad_packed_sequence(target, batch_first=True)
            lens = [int(n) for n in lens]
        B = input.size(0)
        ones, rests, t_one, t_rest, owner, scale = [], [], [], [], [], []
        for b in range(B):
            n = lens[b]
            tb = target[b, :n]                                   # (n, *)
            total = tb.sum(dim=0, keepdim=True)
            t_one.append(tb)
            t_rest.append(total - tb)                            # sum of the others, for every candidate at once
            ones.append(input[b, 0:1].expand(n, *input.shape[2:]))
            rests.append(input[b, 1:2].expand(n, *input.shape[2:]))
            owner += [b] * n
            scale += [1.0 / (n - 1)] * n
        ones, rests = torch.cat(ones, 0).contiguous(), torch.cat(rests, 0).contiguous()
        t_one, t_rest = torch.cat(t_one, 0).contiguous(), torch.cat(t_rest, 0).contiguous()
        loss_one = self.criterion(ones, t_one, batch_mean=False)
        loss_rest = self.criterion(rests, t_rest, batch_mean=False)
        cand = loss_one + loss_rest * torch.tensor(scale, device=loss_one.device, dtype=loss_one.dtype)
        maximize = bool(getattr(self.criterion, "maximize", False))
        losses, indices, start = [], [], 0
        for b in range(B):
            seg = cand[start:start + lens[b]]
            val, idx = (seg.max(dim=0) if maximize else seg.min(dim=0))
            losses.append(val)
            indices.append(idx)
            start += lens[b]
        batch_loss, batch_indices = torch.stack(losses), torch.stack(indices)
        if batch_mean:
            batch_loss = batch_loss.mean(dim=0)
        return batch_loss, batch_indices


from sepkernels.shadowed import fall_through as _fall_through      # names of the reference's same-named module this tree does not define

__getattr__ = _fall_through(__name__, __file__)
