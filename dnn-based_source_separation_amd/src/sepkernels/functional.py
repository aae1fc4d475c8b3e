"""
Autograd-capable building blocks on top of the C ABI, for models that are not one fused kernel sequence.
Used by DPRNN-TasNet (models/dprnn_tasnet.py): the encoder+gLN+bottleneck "head" and the
PReLU+mask+decoder "tail" are the same kernels as in Conv-TasNet (sepkernels/net.py), exposed as two
torch.autograd.Functions working on padded (B, C, ldt) tensors; Segment1d / OverlapAdd1d are index-map kernels.
DPTNet / GALRNet / SepFormer (models/dptnet.py, galrnet.py, sepformer.py) start and end differently (no gLN in front of the
bottleneck or no bottleneck at all; a gated tanh unit between the mask convolution and the decoder), so the same kernels are
also exposed one operation at a time: EncodeFn, PaddedPointwiseFn, MaskDecodeFn.
"""
import torch

from . import backend, LSTM_INTERLEAVED, STATS_SLOTS
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


class EncodeFn(torch.autograd.Function):
    """Learned analysis basis alone (reference models/filterbank.py:205-235 incl. the input padding of the TasNet
    forwards): mixture (B, Cin, T), basis (N, Cin, L) -> w (B, N, ldt) = [ReLU] conv1d(pad(mixture)), frames >= T' zero.
    For the separators whose first operation is not the gLN + bottleneck pair HeadFn fuses (DPTNet, GALRNet)."""

    @staticmethod
    def forward(ctx, mixture, weight, stride, relu):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient w.r.t. the input mixture is not implemented")
        K = backend()
        mixture = mixture.contiguous()
        B, Cin, T_in = mixture.shape
        N, L = weight.shape[0], weight.shape[2]
        geo = _net.Geometry(T_in, L, stride)
        w = torch.empty(B, N, geo.ldt, device=mixture.device, dtype=mixture.dtype)
        stats = torch.zeros(B, STATS_SLOTS, 2, device=mixture.device, dtype=torch.float64)     # the kernel's by-product, unused here
        K.encoder_fwd(mixture, weight, w, stats, B, Cin, T_in, N, L, stride, geo.F, geo.ldt, geo.pad_left, bool(relu))
        ctx.save_for_backward(mixture, w if relu else None)
        ctx.meta = (B, Cin, T_in, N, L, stride, geo, bool(relu))
        return w

    @staticmethod
    def backward(ctx, dw):
        K = backend()
        mixture, w = ctx.saved_tensors
        B, Cin, T_in, N, L, S, geo, relu = ctx.meta
        f32 = dict(device=dw.device, dtype=dw.dtype)
        dpre = torch.where(w > 0, dw, torch.zeros_like(dw)) if relu else dw
        Fx = torch.empty(B, Cin * L, geo.ldt, **f32)
        K.unfold(mixture, Fx, B, Cin, T_in, L, S, geo.F, geo.ldt, geo.pad_left)
        part, _, ns = _net._wgrad(K, B, geo.F, geo.ldt, 0.0, f32, N, Cin * L, dpre.contiguous(), Fx, False)
        dE = torch.empty(N, Cin, L, **f32)
        K.reduce_slabs([(part, 0, dE, N * Cin * L, ns, N * Cin * L, 0, 1.0)])
        return None, dE, None, None


class PaddedPointwiseFn(torch.autograd.Function):
    """[PReLU ->] nn.Conv1d(kernel_size=1) on rows that already carry the workspace stride: x (B, Cin, ldt) with n_frames
    valid frames -> (B, Cout, ldt), frames beyond zero.  `alpha` (the single PReLU slope) or None.  Same kernels as
    PointwiseConv1dFn minus the repacking; with the slope it is the mask convolution of the TasNet tails without its
    sigmoid (sepkernels/net.py tail_forward / tail_backward)."""

    @staticmethod
    def forward(ctx, x, n_frames, weight, bias, alpha, a_amax=None):
        """a_amax: (1,) device tensor >= max|w| over the weights of this and the backward product (SEP_ARITH_F16X3's operand bound), or None:
        the binding then forms it per call (two small reductions); a model that makes dozens of these calls per pass hands over ONE bound"""
        K = backend()
        x = x.contiguous()
        B, Cin, ldt = x.shape
        Cout = weight.shape[0]
        if Cin % 16 or Cout % 16:
            raise NotImplementedError("PaddedPointwiseFn: channel counts must be multiples of 16 (got {} -> {})".format(Cin, Cout))
        y = torch.empty(B, Cout, ldt, device=x.device, dtype=x.dtype)
        pro = dict(pro_mode=_net.PRO_PRELU, pro_alpha=alpha) if alpha is not None else {}
        K.pw_gemm(B=B, M=Cout, K=Cin, T=n_frames, ldt=ldt, A=weight, X=x, Y=y, bias=bias, a_amax=a_amax, **pro)
        ctx.save_for_backward(x, weight, alpha, a_amax)
        ctx.meta = (B, Cin, Cout, n_frames, ldt, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        x, weight, alpha, a_amax = ctx.saved_tensors
        B, Cin, Cout, F, ldt, has_bias = ctx.meta
        f32 = dict(device=x.device, dtype=x.dtype)
        dy = dy.contiguous()
        dx = torch.empty(B, Cin, ldt, **f32)
        dalpha = None
        if alpha is None:
            K.pw_gemm(B=B, M=Cin, K=Cout, T=F, ldt=ldt, trans_a=1, A=weight, X=dy, Y=dx, a_amax=a_amax)
            part, pb, ns = _net._wgrad(K, B, F, ldt, 0.0, f32, Cout, Cin, dy, x, True)
        else:
            slot = torch.zeros(1, device=x.device, dtype=torch.float64)
            K.pw_gemm(B=B, M=Cin, K=Cout, T=F, ldt=ldt, trans_a=1, A=weight, X=dy, Y=dx, epi_flags=_net.EPI_PRELU_BWD, epi_aux=x,
                      epi_alpha=alpha, epi_dalpha=slot, a_amax=a_amax)
            part, pb, ns = _net._wgrad(K, B, F, ldt, 0.0, f32, Cout, Cin, dy, x, True, x_mode=_net.PRO_PRELU, x_alpha=alpha)
            dalpha = torch.empty_like(alpha)
            K.f64_to_f32(slot, dalpha, 1, 0)
        dW, db = torch.empty_like(weight), torch.empty(Cout, **f32)
        K.reduce_slabs([(part, 0, dW, Cout * Cin, ns, Cout * Cin, 0, 1.0), (pb, 0, db, Cout, ns, Cout, 0, 1.0)])
        return dx, None, dW, (db if has_bias else None), dalpha, None


class MaskDecodeFn(torch.autograd.Function):
    """est[b, s] = crop(overlap-add(D^T (w[b] * mask[b, s]))): the last two lines of every masking TasNet (reference
    dptnet.py:139-147 and the same lines of galrnet.py / sepformer.py) for an ALREADY ACTIVATED mask.
    w (B, N, ldt), mask (B, n_src*N, ldt), D (N, Cin, L) -> est (B, n_src, Cin, T) [, latent (B, n_src, N, ldt)]."""

    @staticmethod
    def forward(ctx, w, mask, weight, stride, T_in, want_latent):
        K = backend()
        w, mask = w.contiguous(), mask.contiguous()
        B, N, ldt = w.shape
        n_src = mask.shape[1] // N
        Cin, L = weight.shape[1], weight.shape[2]
        geo = _net.Geometry(T_in, L, stride)
        assert geo.ldt == ldt, "w does not carry the workspace stride of this input length"
        f32 = dict(device=w.device, dtype=w.dtype)
        est = torch.empty(B, n_src, Cin, T_in, **f32)
        latent = torch.empty(B, n_src, N, ldt, **f32) if want_latent else None
        K.decoder_fwd(w, mask, weight, est, latent, B, n_src, N, Cin, L, stride, geo.F, ldt, T_in, geo.pad_left)
        ctx.save_for_backward(w, mask, weight)
        ctx.meta = (B, n_src, N, Cin, L, stride, geo, T_in)
        ctx.set_materialize_grads(False)
        if want_latent:
            ctx.mark_non_differentiable(latent)
            return est, latent
        return est

    @staticmethod
    def backward(ctx, d_est, *unused):
        K = backend()
        w, mask, D = ctx.saved_tensors
        B, n_src, N, Cin, L, S, geo, T_in = ctx.meta
        F, ldt = geo.F, geo.ldt
        f32 = dict(device=w.device, dtype=w.dtype)
        d_est = d_est.contiguous()
        Fd = torch.empty(B * n_src, Cin * L, ldt, **f32)
        K.unfold(d_est, Fd, B * n_src, Cin, T_in, L, S, F, ldt, geo.pad_left)
        part, _, ns = _net._wgrad(K, B, F, ldt, 0.0, f32, N, Cin * L, mask, Fd, False, Bq=B * n_src, Gaux=w, g_mul=1, g_div=n_src)
        dD = torch.empty_like(D)
        K.reduce_slabs([(part, 0, dD, N * Cin * L, ns, N * Cin * L, 0, 1.0)])
        dmask = torch.empty(B, n_src * N, ldt, **f32)
        dw = torch.empty(B, N, ldt, **f32)
        K.decoder_bwd(d_est, w, mask, D, dmask, dw, B, n_src, N, Cin, L, S, F, ldt, T_in, geo.pad_left, raw_mask=1)
        return dw, dmask, dD, None, None, None


def cln_workspace(K, B, C, T, ldt, device):
    """the scratch sep_cln_fwd / sep_cln_bwd ask for at this shape (sep_cln_ws_bytes), as fp64 words"""
    return torch.empty((K.cln_ws_bytes(B, C, T, ldt) + 7) // 8, device=device, dtype=torch.float64)


class PaddedCLNFn(torch.autograd.Function):
    """[PReLU ->] CumulativeLayerNorm1d on rows that already carry the workspace stride: x (B, C, ldt) with n_frames valid frames -> the
    same shape, frames beyond zero (reference src/modules/norm.py:58-101, behind nonlinear1d of tdcn.py:113-116 / 182-186 when `alpha` -- the
    single PReLU slope -- is given).  sep_cln_fwd / sep_cln_bwd (csrc/cln.hip): column sums over the channels, fp64 prefix / suffix sums per
    sample, one apply pass; the activation costs no pass of its own."""

    @staticmethod
    def forward(ctx, x, n_frames, alpha, gamma, beta, eps):
        K = backend()
        x = x.contiguous()
        B, C, ldt = x.shape
        f32 = dict(device=x.device, dtype=x.dtype)
        g1, b1 = gamma.reshape(-1).contiguous(), beta.reshape(-1).contiguous()
        y = torch.empty(B, C, ldt, **f32)
        mean, rstd = torch.empty(B, ldt, **f32), torch.empty(B, ldt, **f32)
        ws = cln_workspace(K, B, C, n_frames, ldt, x.device)
        K.cln_fwd(x, g1, b1, y, mean, rstd, ws, B, C, n_frames, ldt, eps, alpha=alpha)
        ctx.save_for_backward(x, g1, mean, rstd, alpha)
        ctx.meta = (B, C, n_frames, ldt, eps, gamma.shape, beta.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        x, g1, mean, rstd, alpha = ctx.saved_tensors
        B, C, F, ldt, eps, gshape, bshape = ctx.meta
        f32 = dict(device=x.device, dtype=x.dtype)
        dx = torch.empty(B, C, ldt, **f32)
        pg, pb = torch.empty(B, C, **f32), torch.empty(B, C, **f32)
        pa = torch.empty(B, C, **f32) if alpha is not None else None
        ws = cln_workspace(K, B, C, F, ldt, x.device)
        K.cln_bwd(dy.contiguous(), x, g1, mean, rstd, dx, pg, pb, ws, B, C, F, ldt, eps, alpha=alpha, dalpha_part=pa)
        dgamma, dbeta = torch.empty(C, **f32), torch.empty(C, **f32)
        segs = [(pg, 0, dgamma, C, B, C, 0, 1.0), (pb, 0, dbeta, C, B, C, 0, 1.0)]
        K.reduce_slabs(segs)
        dalpha = pa.sum(dtype=torch.float64).to(x.dtype).view(alpha.shape) if alpha is not None else None      # B*C row partials -> the one slope
        return dx, None, dalpha, dgamma.view(gshape), dbeta.view(bshape), None


class PaddedHeadsFn(torch.autograd.Function):
    """The two 1x1 heads of a TCN layer on workspace rows in ONE pass over their input (reference tdcn.py:188-196 + :145-147 + the skip sum
    of :36-41, 70-75): out = Wo v + bo + x_res, total += Ws v + bs.  v (B, H, ldt); x_res (B, Bn, ldt); total (B, Sc, ldt) or None (first
    layer: created) -- UPDATED IN PLACE and returned; Wo / bo None for a layer without the output head.  With Bn % 128 == 0 and the two
    weight matrices adjacent in memory (ConvTasNet's flat parameter buffer keeps them so) it is one product over [Wo; Ws] with the residual
    and the accumulation in its epilogue, and one product [Wo; Ws]^T [d out; d total] backward -- instead of two products, two additions
    forward and two products plus the addition of two H-tensors backward."""

    @staticmethod
    def forward(ctx, v, n_frames, Wo, bo, Ws, bs, x_res, total, a_amax=None):
        K = backend()
        v = v.contiguous()
        B, H, ldt = v.shape
        Sc = Ws.shape[0]
        Bn = Wo.shape[0] if Wo is not None else 0
        f32 = dict(device=v.device, dtype=v.dtype)
        first = total is None
        if first:
            total = torch.empty(B, Sc, ldt, **f32)
        else:
            ctx.mark_dirty(total)
        xo = torch.empty(B, Bn, ldt, **f32) if Wo is not None else None
        joint = Wo is not None and bo is not None and bs is not None and Bn % 128 == 0 and _net._adjacent(Wo, Ws) and _net._adjacent(bo, bs)
        if joint:
            K.pw_gemm(B=B, M=Bn + Sc, K=H, T=n_frames, ldt=ldt, A=Wo.as_strided((Bn + Sc, H), (H, 1)), X=v, Y=xo, Y2=total, m_split=Bn,
                      bias=bo.as_strided((Bn + Sc,), (1,)), accumulate=int(not first), epi_flags=_net.EPI_RESIDUAL, epi_res=x_res, a_amax=a_amax)
        else:
            if Wo is not None:
                K.pw_gemm(B=B, M=Bn, K=H, T=n_frames, ldt=ldt, A=Wo, X=v, Y=xo, bias=bo, epi_flags=_net.EPI_RESIDUAL, epi_res=x_res, a_amax=a_amax)
            K.pw_gemm(B=B, M=Sc, K=H, T=n_frames, ldt=ldt, A=Ws, X=v, Y=total, bias=bs, accumulate=int(not first), a_amax=a_amax)
        ctx.save_for_backward(v, Wo, Ws, a_amax)
        ctx.meta = (B, H, Bn, Sc, n_frames, ldt, first)
        ctx.set_materialize_grads(False)
        return (xo, total) if Wo is not None else (None, total)

    @staticmethod
    def backward(ctx, d_out, d_total):
        K = backend()
        v, Wo, Ws, a_amax = ctx.saved_tensors
        B, H, Bn, Sc, F, ldt, first = ctx.meta
        f32 = dict(device=v.device, dtype=v.dtype)
        if d_total is None:
            d_total = torch.zeros(B, Sc, ldt, **f32)
        d_total = d_total.contiguous()
        have_o = Wo is not None and d_out is not None
        if have_o:
            d_out = d_out.contiguous()
        dv = torch.empty(B, H, ldt, **f32)
        if have_o:
            K.pw_gemm(B=B, M=H, K=Bn + Sc, T=F, ldt=ldt, trans_a=1, A=Wo, A2=Ws, X=d_out, X2=d_total, k_split=Bn, Y=dv, a_amax=a_amax)
        else:
            K.pw_gemm(B=B, M=H, K=Sc, T=F, ldt=ldt, trans_a=1, A=Ws, X=d_total, Y=dv, a_amax=a_amax)
        segs = []
        dWo = dbo = None
        dWs, dbs = torch.empty_like(Ws), torch.empty(Sc, **f32)
        if have_o and Bn % 128 == 0:
            dWo, dbo = torch.empty_like(Wo), torch.empty(Bn, **f32)
            part, pb, ns = _net._wgrad(K, B, F, ldt, 0.0, f32, Bn + Sc, H, d_out, v, True, G2=d_total, g_split=Bn)
            segs += [(part, 0, dWo, Bn * H, ns, (Bn + Sc) * H, 0, 1.0), (part, Bn * H, dWs, Sc * H, ns, (Bn + Sc) * H, 0, 1.0),
                     (pb, 0, dbo, Bn, ns, Bn + Sc, 0, 1.0), (pb, Bn, dbs, Sc, ns, Bn + Sc, 0, 1.0)]
        else:
            if have_o:
                dWo, dbo = torch.empty_like(Wo), torch.empty(Bn, **f32)
                part, pb, ns = _net._wgrad(K, B, F, ldt, 0.0, f32, Bn, H, d_out, v, True)
                segs += [(part, 0, dWo, Bn * H, ns, Bn * H, 0, 1.0), (pb, 0, dbo, Bn, ns, Bn, 0, 1.0)]
            part2, pb2, ns2 = _net._wgrad(K, B, F, ldt, 0.0, f32, Sc, H, d_total, v, True)
            segs += [(part2, 0, dWs, Sc * H, ns2, Sc * H, 0, 1.0), (pb2, 0, dbs, Sc, ns2, Sc, 0, 1.0)]
        K.reduce_slabs(segs)
        return dv, None, dWo, dbo, dWs, dbs, (d_out if have_o else None), (None if first else d_total), None


class PaddedDepthwiseFn(torch.autograd.Function):
    """nn.Conv1d(C, C, k, dilation=d, groups=C) of a TCN layer on rows that carry the workspace stride, with the layer's zero padding folded
    in: `left` zeros in front (causal: (k - 1) d, all of it; else the smaller half -- reference tdcn.py:118-129), the output has the input's
    n_frames.  x (B, C, ldt) with frames >= n_frames zero -> (B, C, ldt); sep_depthwise_* (csrc/stream.hip) on rows of ldt frames -- what
    the kernel writes beyond n_frames is whatever the taps reach there and is ignored by every consumer (they all take n_frames)."""

    @staticmethod
    def forward(ctx, x, n_frames, weight, bias, dilation, left):
        K = backend()
        x = x.contiguous()
        B, C, ldt = x.shape
        Kw = weight.shape[-1]
        y = torch.empty(B, C, ldt, device=x.device, dtype=x.dtype)
        K.depthwise_fwd(x, weight, bias, y, B, C, ldt, ldt, Kw, 1, left, dilation)
        ctx.save_for_backward(x, weight)
        ctx.meta = (B, C, ldt, Kw, dilation, left, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        x, weight = ctx.saved_tensors
        B, C, ldt, Kw, dilation, left, has_bias = ctx.meta
        dy = dy.contiguous()                                     # zero beyond n_frames: it comes out of a kernel that writes the pad frames
        f32 = dict(device=x.device, dtype=x.dtype)
        dx = torch.empty(B, C, ldt, **f32)
        K.depthwise_bwd_input(dy, weight, dx, B, C, ldt, ldt, Kw, 1, left, dilation)
        part = torch.empty(B, C, Kw + 1, **f32)
        K.depthwise_bwd_weight(dy, x, part, B, C, ldt, ldt, Kw, 1, left, dilation)
        dwb = torch.empty(C * (Kw + 1), **f32)
        K.reduce_slabs([(part, 0, dwb, C * (Kw + 1), B, C * (Kw + 1), 0, 1.0)])
        dwb = dwb.view(C, Kw + 1)
        return dx, None, dwb[:, :Kw].reshape(C, 1, Kw).contiguous(), (dwb[:, Kw].contiguous() if has_bias else None), None, None


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


def _round_up(a, b):
    return (a + b - 1) // b * b


class PointwiseConv1dFn(torch.autograd.Function):
    """nn.Conv1d(kernel_size=1) on the MFMA GEMM: x (B, Cin, T), weight (Cout, Cin, 1), bias (Cout) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = backend()
        B, Cin, T = x.shape
        Cout = weight.shape[0]
        if Cin % 16:
            raise NotImplementedError("PointwiseConv1dFn: in_channels must be a multiple of 16")
        ldt = _round_up(T, 128)
        f32 = dict(device=x.device, dtype=x.dtype)
        xp = torch.empty(B, Cin, ldt, **f32)
        K.repack(x.contiguous(), T, xp, ldt, B * Cin, T)
        yp = torch.empty(B, Cout, ldt, **f32)
        K.pw_gemm(B=B, M=Cout, K=Cin, T=T, ldt=ldt, A=weight, X=xp, Y=yp, bias=bias)
        y = torch.empty(B, Cout, T, **f32)
        K.repack(yp, ldt, y, T, B * Cout, T)
        ctx.save_for_backward(xp, weight)
        ctx.meta = (B, Cin, Cout, T, ldt, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        xp, weight = ctx.saved_tensors
        B, Cin, Cout, T, ldt, has_bias = ctx.meta
        f32 = dict(device=xp.device, dtype=xp.dtype)
        dyp = torch.empty(B, Cout, ldt, **f32)
        K.repack(dy.contiguous(), T, dyp, ldt, B * Cout, T)
        dx = None
        if ctx.needs_input_grad[0]:
            if Cout % 16 or Cin % 4:
                raise NotImplementedError("PointwiseConv1dFn backward: out_channels must be a multiple of 16")
            dxp = torch.empty(B, Cin, ldt, **f32)
            K.pw_gemm(B=B, M=Cin, K=Cout, T=T, ldt=ldt, trans_a=1, A=weight, X=dyp, Y=dxp)
            dx = torch.empty(B, Cin, T, **f32)
            K.repack(dxp, ldt, dx, T, B * Cin, T)
        ns = _net._nsplit(Cout, Cin, B * (ldt // 32))
        part = torch.empty(ns, Cout, Cin, **f32)
        pb = torch.empty(ns, Cout, **f32)
        K.pw_wgrad(B=B, M=Cout, N=Cin, T=T, ldt=ldt, G=dyp, X=xp, partial=part, partial_bias=pb, nsplit=ns)
        dW = torch.empty_like(weight)
        db = torch.empty(Cout, **f32)
        K.reduce_slabs([(part, 0, dW, Cout * Cin, ns, Cout * Cin, 0, 1.0), (pb, 0, db, Cout, ns, Cout, 0, 1.0)])
        return dx, dW, (db if has_bias else None)


class DepthwiseConv1dFn(torch.autograd.Function):
    """nn.Conv1d(C, C, k, stride, padding, dilation, groups=C): x (B, C, T), weight (C, 1, k), bias (C) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation):
        K = backend()
        x = x.contiguous()
        B, C, Tin = x.shape
        Kw = weight.shape[-1]
        Tout = (Tin + 2 * padding - dilation * (Kw - 1) - 1) // stride + 1
        y = torch.empty(B, C, Tout, device=x.device, dtype=x.dtype)
        K.depthwise_fwd(x, weight, bias, y, B, C, Tin, Tout, Kw, stride, padding, dilation)
        ctx.save_for_backward(x, weight)
        ctx.meta = (B, C, Tin, Tout, Kw, stride, padding, dilation, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        x, weight = ctx.saved_tensors
        B, C, Tin, Tout, Kw, stride, padding, dilation, has_bias = ctx.meta
        dy = dy.contiguous()
        f32 = dict(device=x.device, dtype=x.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, C, Tin, **f32)
            K.depthwise_bwd_input(dy, weight, dx, B, C, Tin, Tout, Kw, stride, padding, dilation)
        part = torch.empty(B, C, Kw + 1, **f32)
        K.depthwise_bwd_weight(dy, x, part, B, C, Tin, Tout, Kw, stride, padding, dilation)
        dwb = torch.empty(C * (Kw + 1), **f32)
        K.reduce_slabs([(part, 0, dwb, C * (Kw + 1), B, C * (Kw + 1), 0, 1.0)])
        dwb = dwb.view(C, Kw + 1)
        return dx, dwb[:, :Kw].reshape(C, 1, Kw).contiguous(), (dwb[:, Kw].contiguous() if has_bias else None), None, None, None


class ChunkToTokensFn(torch.autograd.Function):
    """(B, F, S, K) -> (B*S, K, F) [inter = False: a sequence per (b, s)] or (B*K, S, F) [inter = True: a sequence per (b, k)] -- the
    permute + reshape in front of the dual-path recurrences (reference dprnn.py:73-76,123-126) as one tiled transpose; backward = the
    inverse kernel."""

    @staticmethod
    def forward(ctx, x, inter):
        B, F, S, K = x.shape
        ctx.dims, ctx.inter = (B, F, S, K), bool(inter)
        y = torch.empty((B * K, S, F) if inter else (B * S, K, F), device=x.device, dtype=x.dtype)
        backend().chunk_to_tokens(x.contiguous(), y, B, F, S, K, inter)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, F, S, K = ctx.dims
        dx = torch.empty(B, F, S, K, device=dy.device, dtype=dy.dtype)
        backend().tokens_to_chunk(dy.contiguous(), dx, B, F, S, K, ctx.inter)
        return dx, None


class TokensToChunkFn(torch.autograd.Function):
    """the inverse: token-major rows back to (B, F, S, K), contiguous"""

    @staticmethod
    def forward(ctx, y, dims, inter):
        B, F, S, K = dims
        ctx.dims, ctx.inter = tuple(dims), bool(inter)
        x = torch.empty(B, F, S, K, device=y.device, dtype=y.dtype)
        backend().tokens_to_chunk(y.contiguous(), x, B, F, S, K, inter)
        return x

    @staticmethod
    def backward(ctx, dx):
        B, F, S, K = ctx.dims
        dy = torch.empty((B * K, S, F) if ctx.inter else (B * S, K, F), device=dx.device, dtype=dx.dtype)
        backend().chunk_to_tokens(dx.contiguous(), dy, B, F, S, K, ctx.inter)
        return dy, None, None


def _dense_ok(K_in, N_out):
    """shapes csrc/linear.hip takes (everything the dual-path separators of the reference's recipes use); others go to torch's BLAS"""
    return K_in % 64 == 0 and N_out % 64 == 0


def _wgrad_slabs(ntok, N, K):
    """number of partial sums of sep_linear_bwd_weight: ~512 workgroups, at least 32 tokens per slab"""
    tiles = max(1, (N // (128 if N % 128 == 0 else 64)) * (K // (128 if K % 128 == 0 else 64)))
    return int(max(1, min(512 // tiles, (ntok + 255) // 256)))


def dense_forward(x2, w, bias=None, bias2=None):
    """x2 (ntok, K) @ w (N, K).t() + bias + bias2 -> (ntok, N)"""
    ntok, Kin = x2.shape
    N = w.shape[0]
    if not (_dense_ok(Kin, N) and takes(x2)):
        out = x2 @ w.t()
        for b in (bias, bias2):
            if b is not None:
                out = out + b
        return out
    y = torch.empty(ntok, N, device=x2.device, dtype=x2.dtype)
    backend().linear_fwd(x2.contiguous(), w.contiguous(), bias, bias2, y, ntok, Kin, N)
    return y


def dense_backward_input(dy2, w, out=None):
    """dy2 (ntok, N) @ w (N, K) -> (ntok, K); `out`: added to it in place"""
    ntok, N = dy2.shape
    Kin = w.shape[1]
    if not (_dense_ok(Kin, N) and takes(dy2)):
        r = dy2 @ w
        return r if out is None else out.add_(r)
    dx = out if out is not None else torch.empty(ntok, Kin, device=dy2.device, dtype=dy2.dtype)
    backend().linear_bwd_input(dy2.contiguous(), w.contiguous(), dx, ntok, Kin, N, out is not None)
    return dx


def dense_backward_weights(jobs):
    """jobs: [(dy2 (ntok, N), x2 (ntok, K), want_bias, L, shift)] -> [(dy2.t() @ shifted(x2) (N, K), dy2.sum(0) or None)], the partial sums of
    ALL jobs added in one sep_reduce_slabs launch.  shift = -1 / +1: x2 is read one step earlier / later inside sequences of L steps (zero
    beyond the sequence's ends) -- the h_{t-1} operand of an LSTM's recurrent weight gradient, without materialising it."""
    out, segs = [None] * len(jobs), []
    K_ = None
    for n_, (dy2, x2, want_bias, L, shift) in enumerate(jobs):
        ntok, N = dy2.shape
        Kin = x2.shape[1]
        if not (_dense_ok(Kin, N) and takes(dy2)):
            xs = x2
            if shift:
                xv = x2.reshape(ntok // L, L, Kin)
                xs = torch.zeros_like(xv)
                if shift < 0:
                    xs[:, 1:] = xv[:, :-1]
                else:
                    xs[:, :-1] = xv[:, 1:]
                xs = xs.reshape(ntok, Kin)
            out[n_] = (dy2.t() @ xs, (dy2.sum(dim=0) if want_bias else None))
            continue
        K_ = K_ or backend()
        ns = _wgrad_slabs(ntok, N, Kin)
        part = torch.empty(ns, N, Kin, device=dy2.device, dtype=dy2.dtype)
        pb = torch.empty(ns, N, device=dy2.device, dtype=dy2.dtype) if want_bias else None
        if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
            x2 = x2.contiguous()
        K_.linear_bwd_weight(dy2.contiguous(), x2, x2.stride(0), part, pb, ntok, Kin, N, L, shift, ns)
        dw = torch.empty(N, Kin, device=dy2.device, dtype=dy2.dtype)
        segs.append((part, 0, dw, N * Kin, ns, N * Kin, 0, 1.0))
        db = None
        if want_bias:
            db = torch.empty(N, device=dy2.device, dtype=dy2.dtype)
            segs.append((pb, 0, db, N, ns, N, 0, 1.0))
        out[n_] = (dw, db)
    if segs:
        K_.reduce_slabs(segs)
    return out


def dense_backward_weight(dy2, x2, want_bias=False, L=1, shift=0):
    return dense_backward_weights([(dy2, x2, want_bias, L, shift)])[0]


class DenseFn(torch.autograd.Function):
    """nn.Linear on [..., K] activations through csrc/linear.hip (reference src/models/dprnn.py:96-99,143-146: the fc layers)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, weight)
        ctx.xshape, ctx.has_bias = x.shape, bias is not None
        return dense_forward(x2, weight, bias).reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        d2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        dx = dense_backward_input(d2, weight).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw, db = dense_backward_weight(d2, x2, want_bias=ctx.has_bias)
        return dx, dw, db


def linear_apply(x, fc):
    """x (..., K) through an nn.Linear parameter container"""
    if isinstance(fc, torch.nn.Linear) and _dense_ok(fc.in_features, fc.out_features) and takes(x):
        return DenseFn.apply(x, fc.weight, fc.bias)
    return fc(x)


class LSTMDirectionFn(torch.autograd.Function):
    """One direction of nn.LSTM(batch_first=True) with zero initial state: x (nseq, L, F) -> h (nseq, L, H).

    The recurrence is libsepkernels (`sep_lstm_fwd` / `sep_lstm_bwd`: W_hh register-resident, persistent over the L
    steps); the input projection and the three weight-gradient products are token-major dense layers (csrc/linear.hip where
    the feature counts are multiples of 64, torch's BLAS otherwise).
    reference: src/models/dprnn.py:65-148 (nn.LSTM inside IntraChunkRNN / InterChunkRNN)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, reverse):
        K = backend()
        nseq, L, F = x.shape
        H = w_hh.shape[1]
        x2 = x.reshape(nseq * L, F)
        xg = dense_forward(x2, w_ih, b_ih, b_hh)                          # (nseq*L, 4H)
        h = torch.empty(nseq, L, H, device=x.device, dtype=x.dtype)
        gates = torch.empty(nseq, L, 4 * H, device=x.device, dtype=x.dtype)
        cst = torch.empty(nseq, L, H, device=x.device, dtype=x.dtype)
        K.lstm_fwd(xg, w_hh.contiguous(), h, gates, cst, nseq, L, H, bool(reverse))
        ctx.save_for_backward(x2, w_ih, w_hh, h, gates, cst)
        ctx.reverse, ctx.shape = bool(reverse), (nseq, L, F, H)
        return h

    @staticmethod
    def backward(ctx, dh):
        K = backend()
        x2, w_ih, w_hh, h, gates, cst = ctx.saved_tensors
        nseq, L, F, H = ctx.shape
        dxg = torch.empty(nseq, L, 4 * H, device=dh.device, dtype=dh.dtype)
        K.lstm_bwd(dh.contiguous(), gates, cst, w_hh.contiguous(), dxg, nseq, L, H, ctx.reverse)
        d2 = dxg.reshape(nseq * L, 4 * H)
        # h_{t-1} of every step (zero initial state): h read one step earlier (later, for the reversed direction)
        (dw_ih, db), (dw_hh, _) = dense_backward_weights([(d2, x2, True, 1, 0), (d2, h.reshape(nseq * L, H), False, L, 1 if ctx.reverse else -1)])
        dx = dense_backward_input(d2, w_ih).reshape(nseq, L, F)
        return dx, dw_ih, dw_hh, db, db, None


class LSTMBidirectionalFn(torch.autograd.Function):
    """Both directions of a bi-LSTM in ONE pair of sweeps (`reverse = 2`: grid.y = direction), x (nseq, L, F) -> (nseq, L, 2H).
    A single-direction sweep occupies nseq/16 compute units (32 of 256 at the DPRNN-TasNet shapes), so the two directions
    ride side by side instead of one after the other."""

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        K = backend()
        nseq, L, F = x.shape
        H = w_hh_f.shape[1]
        x2 = x.reshape(nseq * L, F)
        if _dense_ok(F, 4 * H) and takes(x2):
            xg = torch.empty(2, nseq * L, 4 * H, device=x.device, dtype=x.dtype)
            K.linear_fwd(x2.contiguous(), w_ih_f.contiguous(), b_ih_f, b_hh_f, xg[0], nseq * L, F, 4 * H)
            K.linear_fwd(x2.contiguous(), w_ih_r.contiguous(), b_ih_r, b_hh_r, xg[1], nseq * L, F, 4 * H)
        else:
            xg = torch.empty(2, nseq * L, 4 * H, device=x.device, dtype=x.dtype)
            torch.addmm(b_ih_f + b_hh_f, x2, w_ih_f.t(), out=xg[0])
            torch.addmm(b_ih_r + b_hh_r, x2, w_ih_r.t(), out=xg[1])
        w_hh = torch.stack([w_hh_f, w_hh_r]).contiguous()
        # h of both directions as ONE (nseq, L, 2H) tensor, the layout nn.LSTM(bidirectional=True) returns (SEP_LSTM_INTERLEAVED): no
        # torch.cat behind the sweeps, no torch.stack of the incoming gradient in front of the reverse sweeps
        h = torch.empty(nseq, L, 2 * H, device=x.device, dtype=x.dtype)
        gates = torch.empty(2, nseq, L, 4 * H, device=x.device, dtype=x.dtype)
        cst = torch.empty(2, nseq, L, H, device=x.device, dtype=x.dtype)
        K.lstm_fwd(xg, w_hh, h, gates, cst, nseq, L, H, 2 | LSTM_INTERLEAVED)
        ctx.save_for_backward(x2, w_ih_f, w_ih_r, w_hh, h, gates, cst)
        ctx.shape = (nseq, L, F, H)
        return h

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        x2, w_ih_f, w_ih_r, w_hh, h, gates, cst = ctx.saved_tensors
        nseq, L, F, H = ctx.shape
        dxg = torch.empty(2, nseq, L, 4 * H, device=dy.device, dtype=dy.dtype)
        K.lstm_bwd(dy.contiguous(), gates, cst, w_hh, dxg, nseq, L, H, 2 | LSTM_INTERLEAVED)
        d2 = dxg.reshape(2, nseq * L, 4 * H)
        h2 = h.reshape(nseq * L, 2 * H)
        dx = dense_backward_input(d2[0], w_ih_f)
        dx = dense_backward_input(d2[1], w_ih_r, out=dx).reshape(nseq, L, F)
        # h_{t-1} as each direction saw it: its half of h one step earlier (forward) / later (reversed), zero at the sequence's end
        (dw_ih_f, db_f), (dw_ih_r, db_r), (dw_hh_f, _), (dw_hh_r, _) = dense_backward_weights(
            [(d2[0], x2, True, 1, 0), (d2[1], x2, True, 1, 0), (d2[0], h2[:, :H], False, L, -1), (d2[1], h2[:, H:], False, L, 1)])
        return (dx, dw_ih_f, dw_hh_f, db_f, db_f, dw_ih_r, dw_hh_r, db_r, db_r)


def lstm_bidirectional(x, rnn):
    """x (nseq, L, F) through the parameters of an nn.LSTM(num_layers=1, batch_first=True) container -> (nseq, L, D*H)."""
    if not rnn.bidirectional:
        return LSTMDirectionFn.apply(x, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, False)
    return LSTMBidirectionalFn.apply(x, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0,
                                     rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse, rnn.bias_ih_l0_reverse, rnn.bias_hh_l0_reverse)



LSTM_KERNEL_HIDDEN = (16, 32, 64, 128)


def attn_core_ok(x, L, head_dim):
    """should csrc/attn.hip take this attention?  Heads 8 / 16 / 32 wide, fp32 on the device, and sequences of at most 256 steps: the kernels
    go to 320, but at 257 (three query tiles, the last with one query; nine key blocks) torch's memory-efficient kernel is faster -- 634
    against 463 us forward + backward at DPTNet's inter-chunk shape, profiles/r05zo_attention.txt"""
    # (x.shape[0] sequences ride a 16-bit grid dimension in sep_attn_*: longer batches -- DPTNet at stride 1 on a long recording -- keep torch's kernel)
    return L <= 256 and head_dim in (8, 16, 32) and x.shape[0] <= 65535 and takes(x) and (backend().name != "hip" or x.dtype == torch.float32)


class AttnCoreFn(torch.autograd.Function):
    """O = dropout(softmax(Q K^T / sqrt(d))) V on the packed projection qkv (N, L, 3, H, d) -> (N, L, H * d): the core of nn.MultiheadAttention
    (torch/nn/functional.py multi_head_attention_forward; reference models/dptnet.py:505-527, galr.py:160-226, sepformer.py's encoder layers)
    on sep_attn_fwd / sep_attn_bwd -- no transposes on either side, no L x L tensor, dropout decided by a hash of (seed, n, h, q, key)."""

    @staticmethod
    def forward(ctx, qkv, p_drop):
        K = backend()
        qkv = qkv.contiguous()
        N, L, three, H, D = qkv.shape
        o = torch.empty(N, L, H, D, device=qkv.device, dtype=qkv.dtype)
        lse = torch.empty(N, H, L, device=qkv.device, dtype=qkv.dtype)
        seed = _dropout_seed(qkv) if p_drop > 0 else 0
        scale = float(D) ** -0.5
        K.attn_fwd(qkv, o, lse, N, L, H, D, scale, float(p_drop), seed)
        ctx.save_for_backward(qkv, o, lse)
        ctx.meta = (scale, float(p_drop), seed)
        return o.view(N, L, H * D)

    @staticmethod
    def backward(ctx, dout):
        K = backend()
        qkv, o, lse = ctx.saved_tensors
        N, L, three, H, D = qkv.shape
        scale, p_drop, seed = ctx.meta
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        K.attn_bwd(qkv, o, dout.contiguous(), lse, delta, dqkv, N, L, H, D, scale, p_drop, seed)
        return dqkv, None


def attention_core(qkv, p_drop=0.0):
    """qkv (N, L, 3, H, d) -> (N, L, H * d): csrc/attn.hip where it applies, torch's scaled_dot_product_attention otherwise"""
    N, L, three, H, D = qkv.shape
    if attn_core_ok(qkv, L, D):
        return AttnCoreFn.apply(qkv, float(p_drop))
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    return torch.nn.functional.scaled_dot_product_attention(q, k, v, dropout_p=float(p_drop)).transpose(1, 2).reshape(N, L, H * D)


def _gln_tokens_ws(K, nseq, L, C, device):
    """the scratch sep_gln_tokens_* ask for at this shape (few long sequences are cut into slices), or None"""
    nbytes = K.gln_tokens_ws_bytes(nseq, L, C)
    return torch.empty((nbytes + 7) // 8, device=device, dtype=torch.float64) if nbytes else None


class TokenGLNFn(torch.autograd.Function):
    """Global layer norm of token-major rows: x (nseq, L, C), features contiguous -> the same shape; statistics over the L * C values of a
    sequence, gain / shift per feature (GlobalLayerNorm applied to x.permute(0, 2, 1): what the dual-path transformer blocks of the reference
    do around every attention / feed-forward sub-block, dptnet.py:505-560) without the two transposing copies.  sep_gln_tokens_fwd / bwd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        K = backend()
        x = x.contiguous()
        nseq, L, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(nseq, 2, device=x.device, dtype=x.dtype)
        K.gln_tokens_fwd(x, gamma, beta, y, stats, nseq, L, C, eps, ws=_gln_tokens_ws(K, nseq, L, C, x.device))
        ctx.save_for_backward(x, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        x, gamma, stats = ctx.saved_tensors
        nseq, L, C = x.shape
        dx = torch.empty_like(x)
        part = torch.empty(nseq, 2, C, device=x.device, dtype=x.dtype)
        K.gln_tokens_bwd(dy.contiguous(), x, gamma, stats, dx, part, nseq, L, C, ws=_gln_tokens_ws(K, nseq, L, C, x.device))
        tot = part.sum(0)
        return dx, tot[0], tot[1], None


_seed_rank = {"rank": None, "calls": 0}


def _mix64(x):
    """splitmix64 finaliser: a 64-bit hash of a 64-bit integer"""
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def _dropout_seed(t):
    """a 62-bit seed for a hash-masked dropout (sep_attn_*, sep_rownorm_*, sep_relu_drop_*).  For a tensor on the GPU it is a hash of the
    device generator's (seed, offset), and the offset is advanced -- what torch's own dropout kernels do with that generator: the masks
    are a function of torch.manual_seed(), NOTHING is drawn from the global CPU generator (which the data loaders' shuffling shares with the
    reference) and no .item() costs a launch-bound step host time.  CPU tensors (the tests' stand-in) draw from the CPU generator.
    The rank (cached) and the device are folded in -- data-parallel ranks seeded alike must not drop the same elements."""
    st = _seed_rank
    if st["rank"] is None or st["calls"] % 4096 == 0:
        st["rank"] = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    st["calls"] += 1
    if t.is_cuda:
        gen = torch.cuda.default_generators[t.device.index if t.device.index is not None else torch.cuda.current_device()]
        base, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
        seed = _mix64((base & 0xFFFFFFFFFFFFFFFF) ^ _mix64(off))
    else:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return (seed ^ (0x9E3779B97F4A7C15 * (1 + st["rank"] + 1024 * (t.device.index or 0)))) & (2 ** 62 - 1)


def _contiguous16(t):
    """contiguous AND 16-byte aligned (an incoming gradient may be an offset view of a larger buffer): the float4 kernels' operand form"""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _aligned16(t):
    """float4 accesses: the kernels of csrc/rownorm.hip read rows from the tensor's first byte (a contiguous view that starts inside another
    tensor's row may not be 16-byte aligned; such a caller keeps torch's kernels)"""
    return (t.data_ptr() if t.is_contiguous() else 0) % 16 == 0


class RowNormFn(torch.autograd.Function):
    """y = LayerNorm_C(x + dropout(res)) on token-major rows (..., C): the tail of both sub-blocks of a post-norm nn.TransformerEncoderLayer
    (torch/nn/modules/transformer.py `norm1(x + _sa_block(x))`, `norm2(x + _ff_block(x))`; reference models/sepformer.py:395-520), one pass each
    way on sep_rownorm_fwd / bwd instead of dropout, add, layer norm and their three backward kernels.  res None: y = LayerNorm_C(x)
    (GALRNet's channel norm, models/galr.py:172-190).  The dropout mask is a hash of (seed, element), formed again in the backward pass."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p_drop, keep=True):
        K = backend()
        x = x.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        if res is not None:
            res = res.contiguous()
        p_drop = float(p_drop) if res is not None else 0.0
        seed = _dropout_seed(x) if p_drop > 0 else 0
        # keep: the sum x + dropout(res) is the backward pass's operand -- decided by the CALLER (residual_layer_norm), where the grad mode
        # is still visible: inside Function.forward it is always off and needs_input_grad ignores torch.no_grad()
        keep = res is not None and bool(keep)
        s = torch.empty_like(x) if keep else None
        y = torch.empty_like(x)
        stat = torch.empty(rows, 2, device=x.device, dtype=x.dtype)
        K.rownorm_fwd(x, res, gamma, beta, s, y, stat, rows, C, float(eps), p_drop, seed)
        ctx.save_for_backward(x if res is None else s, gamma, stat)
        ctx.meta = (rows, C, p_drop, seed, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        s, gamma, stat = ctx.saved_tensors
        rows, C, p_drop, seed, has_res = ctx.meta
        ds = torch.empty_like(s)
        dres = torch.empty_like(s) if p_drop > 0 else None
        part = torch.empty(K.rownorm_parts(rows, C), 2, C, device=s.device, dtype=s.dtype)
        if s is None:
            raise RuntimeError("RowNormFn: the forward pass ran without keeping its sum (keep=False) and cannot be differentiated")
        K.rownorm_bwd(_contiguous16(dy), s, gamma, stat, ds, dres, part, rows, C, p_drop, seed)
        tot = part.sum(0)
        return ds, ((dres if dres is not None else ds) if has_res else None), tot[0], tot[1], None, None, None


def rownorm_ok(x, norm):
    """can `norm` (an nn.LayerNorm over the last axis, with gain and shift) run on sep_rownorm_* for rows x (..., C)?"""
    C = x.shape[-1]
    return (isinstance(norm, torch.nn.LayerNorm) and tuple(norm.normalized_shape) == (C,) and norm.weight is not None and norm.bias is not None
            and C % 4 == 0 and 4 <= C <= 1024 and x.numel() > 0 and takes(x) and _aligned16(x) and _aligned16(norm.weight) and _aligned16(norm.bias))


def residual_layer_norm(x, res, norm, p_drop=0.0):
    """norm(x + dropout(res, p_drop)) (res None: norm(x)): sep_rownorm_* where it applies, torch's kernels otherwise"""
    if rownorm_ok(x, norm) and (res is None or (res.shape == x.shape and res.dtype == x.dtype and _aligned16(res))):
        keep = torch.is_grad_enabled() and (x.requires_grad or (res is not None and res.requires_grad) or norm.weight.requires_grad or norm.bias.requires_grad)
        return RowNormFn.apply(x, res, norm.weight, norm.bias, norm.eps, p_drop, keep)
    if res is not None:
        x = x + torch.nn.functional.dropout(res, p_drop, training=p_drop > 0)
    return torch.nn.functional.layer_norm(x, norm.normalized_shape, norm.weight, norm.bias, norm.eps)


class ReluDropFn(torch.autograd.Function):
    """dropout(relu(h), p) between the two Linear layers of a transformer layer's feed-forward sub-block: one pass each way on sep_relu_drop_*
    instead of four torch kernels; the backward pass reads the mask off the forward's output (zero exactly where the gradient is)."""

    @staticmethod
    def forward(ctx, h, p_drop):
        K = backend()
        h = h.contiguous()
        a = torch.empty_like(h)
        K.relu_drop_fwd(h, a, h.numel(), float(p_drop), _dropout_seed(h) if p_drop > 0 else 0)
        ctx.save_for_backward(a)
        ctx.p_drop = float(p_drop)
        return a

    @staticmethod
    def backward(ctx, dy):
        K = backend()
        a, = ctx.saved_tensors
        dh = torch.empty_like(a)
        K.relu_drop_bwd(_contiguous16(dy), a, dh, a.numel(), ctx.p_drop)
        return dh, None


def relu_dropout(h, p_drop):
    """dropout(relu(h), p_drop) -- sep_relu_drop_* where they apply (p_drop = 0: a plain ReLU)"""
    if takes(h) and h.numel() > 0 and h.numel() % 4 == 0 and 0 <= p_drop < 1 and _aligned16(h):
        return ReluDropFn.apply(h, p_drop)
    return torch.nn.functional.dropout(torch.relu(h), p_drop, training=p_drop > 0)


def token_gln_ok(x, norm1d):
    """can `norm1d` (a modules.norm.GlobalLayerNorm) run on token-major rows x (nseq, L, C) through TokenGLNFn?"""
    C = x.shape[-1]
    return (type(norm1d).__name__ == "GlobalLayerNorm" and x.dim() == 3 and C >= 4 and 1024 % C == 0 and x.shape[0] <= 65535 and takes(x)
            and (backend().name != "hip" or x.dtype == torch.float32))      # (nseq rides a 16-bit grid dimension in sep_gln_tokens_*)


def dense_apply(x, weight, bias):
    """x (..., K) @ weight^T + bias with weight (N, K): csrc/linear.hip for the widths it takes, torch's BLAS otherwise"""
    if _dense_ok(weight.shape[1], weight.shape[0]) and takes(x):
        return DenseFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def takes(x):
    """can this tensor go to libsepkernels as it is?  (fp32 on the GPU; the CPU stand-in of the tests takes any real dtype)"""
    if backend().name != "hip":
        return not torch.is_complex(x)
    return x.is_cuda and x.dtype == torch.float32


def lstm_apply(x, rnn):
    """x (nseq, L, F) through a single-layer nn.LSTM parameter container -> (nseq, L, D*H): the sweep kernels when they
    cover the shape (hidden size, fp32 on the device the backend drives), torch's own LSTM otherwise."""
    if isinstance(rnn, torch.nn.LSTM) and rnn.num_layers == 1 and rnn.hidden_size in LSTM_KERNEL_HIDDEN and takes(x):
        return lstm_bidirectional(x.contiguous(), rnn)
    if rnn.batch_first:
        return rnn(x)[0]
    return rnn(x.transpose(0, 1))[0].transpose(0, 1)
