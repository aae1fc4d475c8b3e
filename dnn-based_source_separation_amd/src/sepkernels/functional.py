"""
Autograd-capable building blocks on top of the C ABI, for models that are not one fused kernel sequence.
Used by DPRNN-TasNet (models/dprnn_tasnet.py): the encoder+gLN+bottleneck "head" and the
PReLU+mask+decoder "tail" are the same kernels as in Conv-TasNet (sepkernels/net.py), exposed as two
torch.autograd.Functions working on padded (B, C, ldt) tensors; Segment1d / OverlapAdd1d are index-map kernels.
"""
import torch

from . import backend, STATS_SLOTS
from . import net as _net

HEAD_KEYS = ("encoder.conv1d.weight", "separator.norm1d.norm.weight", "separator.norm1d.norm.bias",
             "separator.bottleneck_conv1d.weight", "separator.bottleneck_conv1d.bias")
TAIL_KEYS = ("separator.prelu.weight", "separator.mask_conv1d.weight", "separator.mask_conv1d.bias",
             "decoder.conv_transpose1d.weight")


class HeadFn(torch.autograd.Function):
    """mixture (B, Cin, T), 5 head parameters -> (w (B, N, ldt), x0 (B, Bn, ldt)); frames >= T' are zero."""

    @staticmethod
    def forward(ctx, mixture, cfg, *params):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient w.r.t. the input mixture is not implemented")
        P = dict(zip(HEAD_KEYS, params))
        stats0 = torch.zeros(mixture.shape[0], STATS_SLOTS, 2, device=mixture.device, dtype=torch.float64)
        geo, w, x0 = _net.head_forward(cfg, P, mixture, stats0)
        ctx.cfg, ctx.geo = cfg, geo
        ctx.save_for_backward(mixture, stats0, w, *params)
        ctx.set_materialize_grads(False)
        return w, x0

    @staticmethod
    def backward(ctx, dw, dx0):
        mixture, stats0, w = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        P = dict(zip(HEAD_KEYS, params))
        if dx0 is None:
            dx0 = torch.zeros(w.shape[0], ctx.cfg["sep_bottleneck_channels"], w.shape[2], device=w.device, dtype=w.dtype)
        if dw is None:
            dw = torch.zeros_like(w)
        G = {k: torch.empty_like(p) for k, p in P.items()}
        _net.head_backward(ctx.cfg, P, ctx.geo, stats0, w, mixture, dx0.contiguous(), dw.contiguous(), G)
        return (None, None) + tuple(G[k] for k in HEAD_KEYS)


class TailFn(torch.autograd.Function):
    """w (B, N, ldt), core (B, C, ldt), 4 tail parameters -> est (B, n_src, Cin, T) [, latent (B, n_src, N, ldt)]."""

    @staticmethod
    def forward(ctx, w, core, cfg, geo, mixture_shape, want_latent, *params):
        P = dict(zip(TAIL_KEYS, params))
        core = core.contiguous()
        est, latent, m = _net.tail_forward(cfg, P, geo, w, core, mixture_shape, want_latent)
        ctx.cfg, ctx.geo, ctx.mixture_shape = cfg, geo, mixture_shape
        ctx.save_for_backward(w, core, m, *params)
        ctx.set_materialize_grads(False)
        if want_latent:
            ctx.mark_non_differentiable(latent)
            return est, latent
        return est

    @staticmethod
    def backward(ctx, d_est, *unused):
        w, core, m = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        P = dict(zip(TAIL_KEYS, params))
        K = backend()
        G = {k: torch.empty_like(p) for k, p in P.items()}
        dalpha = torch.zeros(1, device=w.device, dtype=torch.float64)
        dcore, dwm = _net.tail_backward(ctx.cfg, P, ctx.geo, w, core, m, ctx.mixture_shape, d_est, G, dalpha)
        K.f64_to_f32(dalpha, G["separator.prelu.weight"], 1, 0)
        return (dwm, dcore, None, None, None, None) + tuple(G[k] for k in TAIL_KEYS)


def segment_geometry(T, chunk_size, hop_size):
    padding = (hop_size - (T - chunk_size) % hop_size) % hop_size
    pad_left = padding // 2
    S = (T + padding - chunk_size) // hop_size + 1
    return pad_left, padding - pad_left, S


class SegmentFn(torch.autograd.Function):
    """x (B, C, ldt) with T valid frames -> (B, C, S, chunk) including the reference's zero padding (dprnn_tasnet.py:335-341)."""

    @staticmethod
    def forward(ctx, x, T, chunk, hop):
        K = backend()
        x = x.contiguous()
        B, C, ldt = x.shape
        pad_left, _, S = segment_geometry(T, chunk, hop)
        out = torch.empty(B, C, S, chunk, device=x.device, dtype=x.dtype)
        K.segment(x, out, B * C, T, ldt, S, chunk, hop, pad_left)
        ctx.meta = (B, C, T, ldt, S, chunk, hop, pad_left)
        return out

    @staticmethod
    def backward(ctx, g):
        K = backend()
        B, C, T, ldt, S, chunk, hop, pad_left = ctx.meta
        dx = torch.empty(B, C, ldt, device=g.device, dtype=g.dtype)
        K.overlap_add(g.contiguous(), dx, B * C, T, ldt, S, chunk, hop, pad_left)
        return dx, None, None, None


class OverlapAddFn(torch.autograd.Function):
    """(B, C, S, chunk) -> (B, C, ldt): overlap-add, crop of the padding, T valid frames, zero beyond."""

    @staticmethod
    def forward(ctx, y, T, ldt, hop):
        K = backend()
        y = y.contiguous()
        B, C, S, chunk = y.shape
        pad_left, _, S_exp = segment_geometry(T, chunk, hop)
        assert S == S_exp, "number of chunks does not match the frame count"
        out = torch.empty(B, C, ldt, device=y.device, dtype=y.dtype)
        K.overlap_add(y, out, B * C, T, ldt, S, chunk, hop, pad_left)
        ctx.meta = (B, C, T, ldt, S, chunk, hop, pad_left)
        return out

    @staticmethod
    def backward(ctx, g):
        K = backend()
        B, C, T, ldt, S, chunk, hop, pad_left = ctx.meta
        dy = torch.empty(B, C, S, chunk, device=g.device, dtype=g.dtype)
        K.segment(g.contiguous(), dy, B * C, T, ldt, S, chunk, hop, pad_left)
        return dy, None, None, None
