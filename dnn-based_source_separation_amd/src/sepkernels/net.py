"""
Host-side orchestration of the fused Conv-TasNet forward/backward on the sepkernels C ABI.

The phase structure is dictated by gLN: its statistic spans a whole sample (C x T'), so every gLN is a
grid-wide dependency.  Per TCN layer (reference src/models/tdcn.py:107-147,177-196) the forward is three
kernels, none of which writes a normalised tensor to HBM:

   F2  a  = W1 x + b1                      (MFMA GEMM; epilogue accumulates stats of PReLU(a))
   DW  z  = dw(gLN(PReLU(a))) + bd         (LDS-staged dilated depthwise; accumulates stats of PReLU(z))
   F3  out = Wo v2 + bo + x ; S += Ws v2   (one MFMA GEMM over [Wo;Ws], v2 = gLN(PReLU(z)) applied on load)

and the backward mirrors it (dgrad GEMM -> row-sum finalize -> depthwise^T -> finalize -> dgrad GEMM, plus two
split-K weight-gradient GEMMs).  Only the pre-activations a, z and the layer input x are kept for backward.

Everything here is plumbing: buffer allocation through torch, descriptor filling, launch order.
"""

import os

import torch

import sepkernels

from . import (ARRIVE_INTS, STATS_SLOTS, EPI_PRELU_BWD, EPI_RESIDUAL, EPI_ROWSUMS, EPI_ROWSUMS_PRELU, EPI_SIGMOID, EPI_STATS_PRELU, PRO_GLN,
               PRO_GLN_BWD, PRO_GLN_PRELU, PRO_PRELU, backend)


def round_up(a, b):
    return (a + b - 1) // b * b


class Geometry:
    """Frame geometry of conv_tasnet.py:145-149 (input padding) + the padded row stride of the workspaces."""

    def __init__(self, T_in, L, S):
        self.T_in = T_in
        self.padding = (S - (T_in - L) % S) % S
        self.pad_left = self.padding // 2
        self.pad_right = self.padding - self.pad_left
        self.F = (T_in + self.padding - L) // S + 1
        self.ldt = round_up(self.F, 128)


def layer_names(cfg):
    out = []
    R, X = cfg["sep_num_blocks"], cfg["sep_num_layers"]
    for r in range(R):
        for x in range(X):
            out.append(("separator.tdcn.net.{}.net.{}.".format(r, x), 2 ** x, not (r == R - 1 and x == X - 1)))
    return out


def check_supported(cfg):
    """Raises NotImplementedError (with the reasons) for configurations outside the family the fused HIP kernel sequence
    implements.  ConvTasNet calls it once, in its constructor, and runs such configurations as the module-by-module
    composition instead; the message is kept on the model as `fused_reason`."""
    problems = []
    if cfg.get("enc_basis") != "trainable" or cfg.get("dec_basis") != "trainable":
        problems.append("enc_basis/dec_basis must be 'trainable'")
    if cfg.get("enc_nonlinear") not in (None, "", "relu"):
        problems.append("enc_nonlinear must be None or 'relu'")
    if cfg.get("causal"):
        problems.append("causal=True (cLN) is not implemented")
    if not cfg.get("separable", True) or not cfg.get("dilated", True):
        problems.append("separable=True and dilated=True are required")
    if cfg.get("sep_nonlinear") != "prelu" or not cfg.get("sep_norm", True):
        problems.append("sep_nonlinear='prelu' and sep_norm=True are required")
    if cfg.get("mask_nonlinear") not in ("sigmoid", "softmax"):
        problems.append("mask_nonlinear must be 'sigmoid' or 'softmax'")
    if cfg.get("sep_kernel_size") != 3:
        problems.append("sep_kernel_size must be 3")
    for k in ("n_basis", "sep_hidden_channels", "sep_bottleneck_channels", "sep_skip_channels"):
        if cfg[k] % 16:
            problems.append("{} must be a multiple of 16".format(k))
    if (cfg["n_sources"] * cfg["n_basis"]) % 16:
        problems.append("n_sources*n_basis must be a multiple of 16")
    if cfg["kernel_size"] % cfg["stride"]:
        problems.append("kernel_size must be divisible by stride")
    if problems:
        raise NotImplementedError("sepkernels fused Conv-TasNet path: " + "; ".join(problems))


def _adjacent(t1, t2):
    return t1.data_ptr() + t1.numel() * t1.element_size() == t2.data_ptr()


def _nsplit(M, N, chunks_total, target_blocks=512):
    ntiles = ((M + 127) // 128) * ((N + 127) // 128)
    return max(1, min(chunks_total, max(1, target_blocks // ntiles)))


class Saved:
    """Activations kept between forward and backward (plain attribute bag)."""
    pass


def _zeros(K, *shape, **kw):
    """torch.zeros through the backend: while a launch sequence is being recorded (sepkernels.Sequence) the clearing has to be one of its ops"""
    z = getattr(K, "zeros", None)
    return z(*shape, **kw) if z is not None else torch.zeros(*shape, **kw)


def amax_over(ts):
    """(1,) tensor >= max|t| over the tensors ts.  Tensors that are views of ONE buffer take one reduction over the span they cover
    (alignment gaps included -- they are zero); a model whose parameters live in one flat buffer plus a few derived tensors (the bases of a
    Fourier / pseudo-inverse filterbank) takes one reduction per buffer, not one per tensor (that was ~300 launches per pass)."""
    groups = {}
    for t in ts:
        groups.setdefault((t.untyped_storage().data_ptr(), t.dtype), []).append(t)
    parts = []
    for g in groups.values():
        lo = min(g, key=lambda t: t.data_ptr())
        hi = max(g, key=lambda t: t.data_ptr())
        n = (hi.data_ptr() - lo.data_ptr()) // lo.element_size() + hi.numel()
        if lo.is_contiguous() and n <= 4 * sum(t.numel() for t in g):
            parts.append(lo.detach().as_strided((n,), (1,)).abs().amax().float())
        else:
            parts.extend(t.detach().abs().amax().float() for t in g)
    return (parts[0] if len(parts) == 1 else torch.stack(parts).amax()).reshape(1)


def _weights_amax(P):
    """(1,) device tensor >= max|w| over all parameters in P: the A-operand bound of SEP_ARITH_F16X3 (None when another
    arithmetic or the CPU emulator is in use)."""
    if sepkernels.gemm_arith() != sepkernels.ARITH_F16X3 or getattr(backend(), "name", "") != "hip":
        return None
    return amax_over([t for t in P.values() if torch.is_tensor(t)])


def pack_weights(cfg, P, need_bwd):
    """Every 1x1-convolution weight of the step, split once for SEP_ARITH_F16X3 (sep_pack_weights: {hi, lo} fp16 groups,
    one power-of-two scale per row) in the orientation of each product: forward products use W, input-gradient products
    W^T, and the two heads of a layer -- adjacent in the flat parameter buffer -- are one matrix [Wo;Ws] either way.
    Returns {} when another arithmetic is selected (the kernels then split the fp32 weights themselves).  Done at the start of
    every forward pass: the weights change between steps through raw pointers (fused Adam), so there is nothing to cache
    on, and the whole pack is ~40 MB of writes (~15 us)."""
    K = backend()
    if sepkernels.gemm_arith() != sepkernels.ARITH_F16X3 or not hasattr(K, "pack_weights"):
        return {}
    N, n_src = cfg["n_basis"], cfg["n_sources"]
    Bn, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    names, specs = [], []

    def add(name, W, r, c):
        names.append(name)
        specs.append((W, r, c, 0))
        if need_bwd:
            names.append(name + "^T")
            specs.append((W, r, c, 1))

    add("bottleneck", P["separator.bottleneck_conv1d.weight"], Bn, N)
    for li, (pre, _, dual) in enumerate(layer_names(cfg)):
        sp = pre + "separable_conv1d."
        add("conv1.{}".format(li), P[pre + "bottleneck_conv1d.weight"], H, Bn)
        Ws = P[sp + "skip_pointwise_conv1d.weight"]
        if dual:
            Wo = P[sp + "output_pointwise_conv1d.weight"]
            if Bn % 128 == 0 and _adjacent(Wo, Ws):
                add("heads.{}".format(li), Wo.as_strided((Bn + Sc, H), (H, 1)), Bn + Sc, H)
            else:
                add("out.{}".format(li), Wo, Bn, H)
                add("skip.{}".format(li), Ws, Sc, H)
        else:
            add("skip.{}".format(li), Ws, Sc, H)
    add("mask", P["separator.mask_conv1d.weight"], n_src * N, Sc)
    # what sep_pack_weights takes: contraction length in 16-column groups, output rows in 32-row blocks (widths like 48 or 80 -- multiples
    # of 16 the model accepts -- are not: those products hand the fp32 weights to sep_pw_gemm, which splits them itself)
    keep = [i for i, (_, r, c, t) in enumerate(specs) if (r if t else c) % 16 == 0 and (c if t else r) % 32 == 0]
    packs = K.pack_weights([specs[i] for i in keep])
    return {names[i]: pk for i, pk in zip(keep, packs)}


def head_forward(cfg, P, mixture, stats0, PK=None):
    """Encoder (+input padding) and the separator's first gLN + 1x1 bottleneck.
    Returns (geo, w (B,N,ldt), x0 (B,Bn,ldt)); stats0 (B,SLOTS,2) zeroed by the caller receives the statistics of w."""
    K = backend()
    B, Cin, T_in = mixture.shape
    N, L, S = cfg["n_basis"], cfg["kernel_size"], cfg["stride"]
    Bn = cfg["sep_bottleneck_channels"]
    eps = float(cfg.get("eps", 1e-12))
    relu = cfg.get("enc_nonlinear") == "relu"
    geo = Geometry(T_in, L, S)
    F, ldt = geo.F, geo.ldt
    f32 = dict(device=mixture.device, dtype=mixture.dtype)
    w = torch.empty(B, N, ldt, **f32)
    K.encoder_fwd(mixture, P["encoder.conv1d.weight"], w, stats0, B, Cin, T_in, N, L, S, F, ldt, geo.pad_left, relu)
    x = torch.empty(B, Bn, ldt, **f32)
    K.pw_gemm(B=B, M=Bn, K=N, T=F, ldt=ldt, A=P["separator.bottleneck_conv1d.weight"], A_pk=(PK or {}).get("bottleneck"), X=w, Y=x,
              bias=P["separator.bottleneck_conv1d.bias"], pro_mode=PRO_GLN, pro_stats=stats0,
              pro_gamma=P["separator.norm1d.norm.weight"], pro_beta=P["separator.norm1d.norm.bias"], count=N * F, eps=eps)
    return geo, w, x


def tail_forward(cfg, P, geo, w, core, mixture_shape, want_latent, PK=None):
    """PReLU -> 1x1 mask conv -> sigmoid | channel softmax -> mask*w -> decoder (overlap-add) -> crop.  core (B, C, ldt)."""
    K = backend()
    B, Cin, T_in = mixture_shape
    N, L, S, n_src = cfg["n_basis"], cfg["kernel_size"], cfg["stride"], cfg["n_sources"]
    eps = float(cfg.get("eps", 1e-12))
    F, ldt = geo.F, geo.ldt
    C = core.shape[1]
    f32 = dict(device=w.device, dtype=w.dtype)
    m = torch.empty(B, n_src * N, ldt, **f32)
    softmax = cfg.get("mask_nonlinear") == "softmax"
    K.pw_gemm(B=B, M=n_src * N, K=C, T=F, ldt=ldt, A=P["separator.mask_conv1d.weight"], A_pk=(PK or {}).get("mask"), X=core, Y=m,
              bias=P["separator.mask_conv1d.bias"], pro_mode=PRO_PRELU, pro_alpha=P["separator.prelu.weight"],
              epi_flags=0 if softmax else EPI_SIGMOID, eps=eps)
    if softmax:      # nn.Softmax(dim=1) over all n_src*N channels of a frame (reference conv_tasnet.py:357)
        K.softmax_ch_fwd(m, B, n_src * N, F, ldt)
    est = torch.empty(B, n_src, Cin, T_in, **f32)
    latent = torch.empty(B, n_src, N, ldt, **f32) if want_latent else None
    K.decoder_fwd(w, m, P["decoder.conv_transpose1d.weight"], est, latent, B, n_src, N, Cin, L, S, F, ldt, T_in, geo.pad_left)
    return est, latent, m


def _nsplit_aligned(M, N, Bq, ldt, target_blocks=512):
    """slab count B * k with k a divisor of ldt / 32: every slab then holds frames of ONE sample (slab s -> sample s // k), whichever
    chunk size the weight-gradient kernel works in (sep_gln_bwd_from_wgrad needs that); as close to _nsplit's count as that allows"""
    ntiles = ((M + 127) // 128) * ((N + 127) // 128)
    want = max(1, target_blocks // ntiles)
    cps = ldt // 32
    k = max([q for q in range(1, cps + 1) if cps % q == 0 and Bq * q <= want] or [1])
    return Bq * k


def _wgrad(K, B, F, ldt, eps, f32, M, Nn, Gt, Xt, want_bias, Bq=None, weps=None, aligned=False, **kw):
    """One weight-gradient product -> (partial slabs, partial bias slabs, number of slabs): `ns` slabs, summed afterwards in a fixed
    order (sep_reduce_slabs, or sep_gln_bwd_from_wgrad for sample-aligned slabs).  Adding every slab onto ONE with fp32 atomics
    instead (sep_wgrad_desc.accumulate) was measured on MI355X and is slower -- 100 vs 82 us per launch, 17.85 vs 17.29 ms per step
    (profiles/r03a_wgrad_atomic.txt): the atomics resolve at the memory side across the eight XCDs."""
    Bq = B if Bq is None else Bq
    ch = Bq * (ldt // 32)
    ns = _nsplit_aligned(M, Nn, Bq, ldt) if aligned else _nsplit(M, Nn, ch)
    part = torch.empty(ns, M, Nn, **f32)
    pb = torch.empty(ns, M, **f32) if want_bias else None
    K.pw_wgrad(B=Bq, M=M, N=Nn, T=F, ldt=ldt, G=Gt, X=Xt, partial=part, partial_bias=pb, nsplit=ns,
               eps=(eps if weps is None else weps), **kw)
    return part, pb, ns


def tail_backward(cfg, P, geo, w, core, m, mixture_shape, d_est, G, dalpha_slot, PK=None):
    """Backward of tail_forward.  Writes the gradients of decoder / mask conv; accumulates the PReLU slope gradient
    into dalpha_slot (double, 1 element, zeroed by the caller).  Returns (dcore (B,C,ldt), dwm (B,N,ldt))."""
    K = backend()
    B, Cin, T_in = mixture_shape
    N, L, S, n_src = cfg["n_basis"], cfg["kernel_size"], cfg["stride"], cfg["n_sources"]
    eps = float(cfg.get("eps", 1e-12))
    F, ldt = geo.F, geo.ldt
    C = core.shape[1]
    f32 = dict(device=w.device, dtype=w.dtype)
    d_est = d_est.contiguous()
    D = P["decoder.conv_transpose1d.weight"]
    Fd = torch.empty(B * n_src, Cin * L, ldt, **f32)
    K.unfold(d_est, Fd, B * n_src, Cin, T_in, L, S, F, ldt, geo.pad_left)
    part, _, ns = _wgrad(K, B, F, ldt, eps, f32, N, Cin * L, m, Fd, False, Bq=B * n_src, Gaux=w, g_mul=1, g_div=n_src)
    K.reduce_slabs([(part, 0, G["decoder.conv_transpose1d.weight"], N * Cin * L, ns, N * Cin * L, 0, 1.0)])
    dpre = torch.empty(B, n_src * N, ldt, **f32)
    dwm = torch.empty(B, N, ldt, **f32)
    softmax = cfg.get("mask_nonlinear") == "softmax"
    K.decoder_bwd(d_est, w, m, D, dpre, dwm, B, n_src, N, Cin, L, S, F, ldt, T_in, geo.pad_left, raw_mask=int(softmax))
    if softmax:
        K.softmax_ch_bwd(m, dpre, B, n_src * N, F, ldt)
    Wm = P["separator.mask_conv1d.weight"]
    alpha_m = P["separator.prelu.weight"]
    dcore = torch.empty(B, C, ldt, **f32)
    K.pw_gemm(B=B, M=C, K=n_src * N, T=F, ldt=ldt, trans_a=1, A=Wm, A_pk=(PK or {}).get("mask^T"), X=dpre, Y=dcore, epi_flags=EPI_PRELU_BWD, epi_aux=core,
              epi_alpha=alpha_m, epi_dalpha=dalpha_slot, eps=eps)
    part, pb, ns = _wgrad(K, B, F, ldt, eps, f32, n_src * N, C, dpre, core, True, x_mode=PRO_PRELU, x_alpha=alpha_m)
    K.reduce_slabs([(part, 0, G["separator.mask_conv1d.weight"], n_src * N * C, ns, n_src * N * C, 0, 1.0),
                    (pb, 0, G["separator.mask_conv1d.bias"], n_src * N, ns, n_src * N, 0, 1.0)])
    return dcore, dwm


def head_backward(cfg, P, geo, stats0, w, mixture, dx0, dwm, G, PK=None, want_dmix=False):
    """Backward of head_forward given dx0 = d(bottleneck output) and dwm = d(w) arriving through mask*w.
    want_dmix: also return d(mixture) (B, Cin, T) -- the adjoint of the analysis convolution is the overlap-add the decoder kernel
    performs (sep_decoder_fwd with the analysis basis and a mask of ones), one extra launch for the rare caller that differentiates
    with respect to the input (reference: autograd through nn.Conv1d, src/models/filterbank.py:212,222-230)."""
    K = backend()
    B, Cin, T_in = mixture.shape
    N, L, S = cfg["n_basis"], cfg["kernel_size"], cfg["stride"]
    Bn = cfg["sep_bottleneck_channels"]
    eps = float(cfg.get("eps", 1e-12))
    relu = cfg.get("enc_nonlinear") == "relu"
    F, ldt = geo.F, geo.ldt
    f32 = dict(device=w.device, dtype=w.dtype)
    nt64 = ldt // 64
    g0, b0 = P["separator.norm1d.norm.weight"], P["separator.norm1d.norm.bias"]
    Wb = P["separator.bottleneck_conv1d.weight"]
    cnt0 = N * F
    part, pb, ns = _wgrad(K, B, F, ldt, eps, f32, Bn, N, dx0, w, True, x_mode=PRO_GLN, x_stats=stats0, x_gamma=g0, x_beta=b0, count=cnt0)
    dvw = torch.empty(B, N, ldt, **f32)
    rp0 = torch.empty(B, N, nt64, 2, **f32)
    K.pw_gemm(B=B, M=N, K=Bn, T=F, ldt=ldt, trans_a=1, A=Wb, A_pk=(PK or {}).get("bottleneck^T"), X=dx0, Y=dvw, epi_flags=EPI_ROWSUMS, epi_aux=w,
              epi_rowpart=rp0, eps=eps)
    bsum0 = torch.empty(B, 2, **f32)
    pbeta0 = torch.empty(B, N, **f32)
    pgamma0 = torch.empty(B, N, **f32)
    K.gln_bwd_finalize(rp0, nt64, 2, stats0, g0, cnt0, eps, bsum0, pbeta0, pgamma0, None, B, N)
    K.head_bwd(dvw, w, dwm, stats0, g0, bsum0, B, N, F, ldt, cnt0, eps, relu)
    K.reduce_slabs([(part, 0, G["separator.bottleneck_conv1d.weight"], Bn * N, ns, Bn * N, 0, 1.0),
                    (pb, 0, G["separator.bottleneck_conv1d.bias"], Bn, ns, Bn, 0, 1.0),
                    (pbeta0, 0, G["separator.norm1d.norm.bias"], N, B, N, 0, 1.0),
                    (pgamma0, 0, G["separator.norm1d.norm.weight"], N, B, N, 0, 1.0)])
    Fx = torch.empty(B, Cin * L, ldt, **f32)
    K.unfold(mixture, Fx, B, Cin, T_in, L, S, F, ldt, geo.pad_left)
    part, _, ns = _wgrad(K, B, F, ldt, eps, f32, N, Cin * L, dvw, Fx, False)
    K.reduce_slabs([(part, 0, G["encoder.conv1d.weight"], N * Cin * L, ns, N * Cin * L, 0, 1.0)])
    if not want_dmix:
        return None
    dmix = torch.empty(B, 1, Cin, T_in, **f32)
    K.decoder_fwd(dvw, torch.ones(B, N, ldt, **f32), P["encoder.conv1d.weight"], dmix, None, B, 1, N, Cin, L, S, F, ldt, T_in, geo.pad_left)
    return dmix.view(B, Cin, T_in)


def forward(cfg, P, mixture, want_latent=False, save=True):
    """See _forward; sets the per-pass weight bound of SEP_ARITH_F16X3 around it."""
    prev = sepkernels.set_weights_amax(_weights_amax(P))
    try:
        return _forward(cfg, P, mixture, want_latent, save)
    finally:
        sepkernels.set_weights_amax(prev)


def _forward(cfg, P, mixture, want_latent, save):
    """cfg: model config dict; P: dict name -> parameter tensor; mixture (B, Cin, T) fp32 contiguous.
    Returns (est (B, n_src, Cin, T), latent or None, Saved or None)."""
    K = backend()
    dev = mixture.device
    B, Cin, T_in = mixture.shape
    N, L, S = cfg["n_basis"], cfg["kernel_size"], cfg["stride"]
    Bn, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    n_src = cfg["n_sources"]
    eps = float(cfg.get("eps", 1e-12))          # separator.norm1d
    teps = float(cfg.get("tcn_eps", 1e-12))     # TCN norms keep tdcn.EPS: the reference never forwards eps to the TCN (conv_tasnet.py:336-339)
    relu = cfg.get("enc_nonlinear") == "relu"
    geo = Geometry(T_in, L, S)
    F, ldt = geo.F, geo.ldt
    layers = layer_names(cfg)
    nl = len(layers)
    f32 = dict(device=dev, dtype=mixture.dtype)   # always fp32 in the product; the CPU emulator tests also run fp64

    stats = _zeros(K, 2 * nl + 1, B, STATS_SLOTS, 2, device=dev, dtype=torch.float64)
    PK = pack_weights(cfg, P, need_bwd=save)
    geo, w, x = head_forward(cfg, P, mixture, stats[0], PK)

    skip = torch.empty(B, Sc, ldt, **f32)
    acts = []
    for li, (pre, dil, dual) in enumerate(layers):
        sp = pre + "separable_conv1d."
        st1, st2 = stats[1 + 2 * li], stats[2 + 2 * li]
        a = torch.empty(B, H, ldt, **f32)
        K.pw_gemm(B=B, M=H, K=Bn, T=F, ldt=ldt, A=P[pre + "bottleneck_conv1d.weight"], A_pk=PK.get("conv1.{}".format(li)), X=x, Y=a,
                  bias=P[pre + "bottleneck_conv1d.bias"], epi_flags=EPI_STATS_PRELU, epi_alpha=P[pre + "nonlinear1d.weight"],
                  epi_stats=st1, eps=teps)
        z = torch.empty(B, H, ldt, **f32)
        K.dwconv_fwd(a, st1, P[pre + "norm1d.norm.weight"], P[pre + "norm1d.norm.bias"], P[pre + "nonlinear1d.weight"],
                     P[sp + "depthwise_conv1d.weight"], P[sp + "depthwise_conv1d.bias"], P[sp + "nonlinear1d.weight"],
                     z, st2, B, H, F, ldt, dil, teps)
        pro = dict(pro_mode=PRO_GLN_PRELU, pro_stats=st2, pro_gamma=P[sp + "norm1d.norm.weight"],
                   pro_beta=P[sp + "norm1d.norm.bias"], pro_alpha=P[sp + "nonlinear1d.weight"], count=H * F, eps=teps)
        Ws, bs = P[sp + "skip_pointwise_conv1d.weight"], P[sp + "skip_pointwise_conv1d.bias"]
        if dual:
            Wo, bo = P[sp + "output_pointwise_conv1d.weight"], P[sp + "output_pointwise_conv1d.bias"]
            xo = torch.empty(B, Bn, ldt, **f32)
            if Bn % 128 == 0 and _adjacent(Wo, Ws) and _adjacent(bo, bs):
                # [Wo;Ws] contiguous (flat parameter layout): one GEMM reads the H-tensor z once for both heads
                Wcat = Wo.as_strided((Bn + Sc, H), (H, 1))      # views over the flat parameter buffer
                bcat = bo.as_strided((Bn + Sc,), (1,))
                K.pw_gemm(B=B, M=Bn + Sc, K=H, T=F, ldt=ldt, A=Wcat, A_pk=PK.get("heads.{}".format(li)), X=z, Y=xo, Y2=skip, m_split=Bn,
                          bias=bcat, accumulate=int(li > 0), epi_flags=EPI_RESIDUAL, epi_res=x, **pro)
            else:
                K.pw_gemm(B=B, M=Bn, K=H, T=F, ldt=ldt, A=Wo, A_pk=PK.get("out.{}".format(li)), X=z, Y=xo, bias=bo, epi_flags=EPI_RESIDUAL,
                          epi_res=x, **pro)
                K.pw_gemm(B=B, M=Sc, K=H, T=F, ldt=ldt, A=Ws, A_pk=PK.get("skip.{}".format(li)), X=z, Y=skip, bias=bs,
                          accumulate=int(li > 0), **pro)
        else:
            xo = None
            K.pw_gemm(B=B, M=Sc, K=H, T=F, ldt=ldt, A=Ws, A_pk=PK.get("skip.{}".format(li)), X=z, Y=skip, bias=bs,
                      accumulate=int(li > 0), **pro)
        if save:
            acts.append((x, a, z))          # inference (torch.no_grad()): nothing is kept, a / z go back to the allocator layer by layer
        x = xo

    est, latent, m = tail_forward(cfg, P, geo, w, skip, mixture.shape, want_latent, PK)

    sv = None
    if save:
        sv = Saved()
        sv.geo, sv.stats, sv.w, sv.acts, sv.skip, sv.m, sv.mixture = geo, stats, w, acts, skip, m, mixture
        sv.packs = PK          # the transposed packs made above are the backward pass's A operands
    return est, latent, sv


class _SideStream:
    """Second HIP stream for the weight-gradient GEMMs of the backward pass.

    The input-gradient chain (G3 -> depthwise^T -> G2 per layer) is serial and alternates MFMA-bound and HBM-bound
    kernels; the weight gradients hang off it as leaves (nothing in the chain reads them).  Launched on a second stream
    they fill the matrix pipe while the chain's streaming kernels wait on HBM, and the small reduction kernels
    disappear under the big ones.  `fork()` orders the side stream after everything enqueued on the main stream so far;
    `join()` orders the main stream after the side stream.  Tensors produced on the main stream and read on the side
    stream are handed to `keep()` so that the caching allocator does not recycle them while the side stream lags.
    Measured on MI355X (paper-best, B=16).  fp32-MFMA arithmetic: no gain -- 30.7 vs 30.5 ms/step: every kernel of that
    path fills the chip's LDS/VGPR slots by itself, so a concurrent kernel only takes slots away from the other one.
    Split arithmetic, round 2: 20.8 vs 21.2 ms/step (same box, twice, bit-identical loss) -- its weight-gradient kernel launched
    512 workgroups onto 768 slots, and the chain's kernels took the rest.  Round 5, with the one-workgroup-per-compute-unit kernels
    of rounds 3 - 4 (nothing of another launch fits beside them): OFF is faster in every A/B pair -- 16.26 vs 16.50 / 16.60,
    16.05 vs 16.39 / 16.41 ms/step at 16 utterances, 10.4 vs 11.7 - 12.4 at 8, where the stream's events also cost host time
    (profiles/r07_round5_experiments.md).  Hence OFF by default; SEPK_SIDE_STREAM=1 turns it on (the tests run both ways); always
    off on CPU tensors (emulator tests)."""
    _streams = {}

    def __init__(self, dev):
        want = os.environ.get("SEPK_SIDE_STREAM", "auto")
        self.on = dev.type == "cuda" and want == "1"
        if self.on:
            key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
            if key not in _SideStream._streams:
                _SideStream._streams[key] = torch.cuda.Stream(device=dev)
            self.side = _SideStream._streams[key]
            self.main = torch.cuda.current_stream(dev)

    def fork(self):
        if self.on:
            self.side.wait_stream(self.main)

    def keep(self, *tensors):
        if self.on:
            for t in tensors:
                if t is not None:
                    t.record_stream(self.side)

    def lend(self, *tensors):
        """the other direction of keep(): tensors ALLOCATED inside `with side:` (side-stream pool) that the main stream reads later
        (queued reduce_slabs segments).  Without the mark the block returns to the side pool when its last reference drops and the next
        side-stream allocation may write it while main's read is still in flight."""
        if self.on:
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)

    def __enter__(self):
        if self.on:
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self._ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)

    def mark(self):
        """an event behind what is queued on the side stream SO FAR"""
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record(self.side)
        return ev

    def wait(self, ev):
        """main waits for a mark() (later side-stream work does not hold main up)"""
        if self.on and ev is not None:
            self.main.wait_event(ev)


def backward(cfg, P, sv, d_est, G, on_ready=None, want_dmix=False):
    """See _backward; sets the per-pass weight bound of SEP_ARITH_F16X3 around it."""
    prev = sepkernels.set_weights_amax(_weights_amax(P))
    try:
        return _backward(cfg, P, sv, d_est, G, on_ready, want_dmix)
    finally:
        sepkernels.set_weights_amax(prev)


def _backward(cfg, P, sv, d_est, G, on_ready, want_dmix=False):
    """Writes the gradient of every parameter into G[name] (overwrites; G tensors have the parameter shapes); returns d(mixture) when
    want_dmix, else None.

    on_ready(first_block, last_block_or_None): optional callback for gradient bucketing.  It is called when every
    gradient of TCN blocks [first_block, ...] is final: first with the last block (plus the whole tail: mask PReLU, mask
    conv, decoder), then once per earlier block, and the caller treats the call for block 0 as "block 0 plus the head"
    by waiting for this function to return.  With a callback the queued reductions are flushed per block."""
    K = backend()
    mixture = sv.mixture
    dev = mixture.device
    B, Cin, T_in = mixture.shape
    N, L, S = cfg["n_basis"], cfg["kernel_size"], cfg["stride"]
    Bn, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    n_src = cfg["n_sources"]
    eps = float(cfg.get("eps", 1e-12))
    teps = float(cfg.get("tcn_eps", 1e-12))
    relu = cfg.get("enc_nonlinear") == "relu"
    geo, stats, w, skip, m = sv.geo, sv.stats, sv.w, sv.skip, sv.m
    F, ldt = geo.F, geo.ldt
    layers = layer_names(cfg)
    nl = len(layers)
    f32 = dict(device=dev, dtype=mixture.dtype)   # always fp32 in the product; the CPU emulator tests also run fp64
    chunks = B * (ldt // 32)
    nt64, nt1024 = ldt // 64, (ldt + 1023) // 1024
    dalpha = _zeros(K, nl + 1, device=dev, dtype=torch.float64)   # [layer alpha1 ..., mask prelu]
    # gLN backward needs two per-sample means of the incoming gradient g -- mean(gamma g), mean(gamma g xhat) -- before any element of the
    # input gradient can be formed.  Inside the TCN layers the kernel that PRODUCES g's sums finishes them: its workgroups add their
    # gamma-weighted totals to fp64 slots (bacc, laid out like `stats`: [1 + 2 li] / [2 + 2 li] = gLN1 / gLN2 of layer li), count their
    # arrivals (arrive), and the sample's last one stores the two means (bsum) the consumer's prologue reads -- no second-stage launch in
    # between (gln_bwd_publish, csrc/common.hpp).  That is gLN2 (16 workgroups per sample in the sums kernel); for gLN1 the depthwise backward --
    # thousands of short workgroups -- only adds to the slots and one thread per workgroup of the conv1^T product forms the means
    # (pro_bacc): the arrival protocol cost that kernel +10 us per launch.  sep_gln_bwd_finalize still turns gLN1's row partials into
    # parameter gradients, as a leaf.
    bacc = _zeros(K, 2 * nl + 1, B, STATS_SLOTS, 2, device=dev, dtype=torch.float64)
    arrive = _zeros(K, 2 * nl + 1, B, ARRIVE_INTS, device=dev, dtype=torch.int32)
    bsum = torch.empty(2 * nl + 1, B, 2, **f32)
    def wgrad(M, Nn, Gt, Xt, dW, dbias=None, Bq=B, weps=None, **kw):
        return _wgrad(K, B, F, ldt, eps, f32, M, Nn, Gt, Xt, dbias is not None, Bq=Bq, weps=weps, **kw)

    # ---- tail: decoder / mask ---------------------------------------------------------------------
    PK = getattr(sv, "packs", None) or {}
    dS, dwm = tail_backward(cfg, P, geo, w, skip, m, mixture.shape, d_est, G, dalpha[nl:nl + 1], PK)

    # The skip gradient dS is the second G source of EVERY layer's heads weight gradient: its half of that kernel's operand split is done once here
    # (sep_split_rows: {hi, lo} fp16 lines, one scale per sample and row, the bias partials of its rows) instead of 24 times inside the kernels'
    # consumer waves (profiles/r08_round6_experiments.md, r08r: 83 -> 70 us per launch).  The fp16 kernel's shape only: [Wo;Ws] as one 256-row product.
    dS_pre = None
    if (hasattr(K, "split_rows") and sepkernels.gemm_arith() == sepkernels.ARITH_F16X3 and Bn == 128 and Sc == 128 and H % 128 == 0
            and os.environ.get("SEPK_WGRAD_PRESPLIT", "1") != "0" and ldt <= 8192):
        k_heads = _nsplit_aligned(Bn + Sc, H, B, ldt) // B
        if (ldt // 32) % k_heads == 0 and k_heads <= 64:
            dS_pre = K.split_rows(dS, F, k_heads)

    # ---- TCN layers, reversed -----------------------------------------------------------------------
    side = _SideStream(dev)
    side.fork()   # dS exists
    # Every second-stage reduction of the layer loop (weight-gradient slabs, per-sample gLN/depthwise partials, PReLU
    # slopes) is a leaf: nothing in the chain reads it.  They are queued and flushed in a handful of 64-segment launches
    # after the loop instead of ~100 tiny launches inside it (each costs a ~5 us dispatch bubble on the critical path).
    # Price: the slabs of all layers stay alive until the flush (~1.6 GB at B=16 paper-best; HBM is 288 GB).
    pending = []
    W1_BATCH = max(1, min(8, int(os.environ.get("SEPK_WGRAD_BATCH", "8"))))      # measured on one box (profiles/r08b_wgrad_batch.txt): 15.81 / 15.60 / 15.42 / 15.40 ms per step at 1 / 2 / 4 / 8
    w1_held = []                   # (da, x, parameter prefix) of layers whose conv1 weight gradient has not been issued yet

    def flush_w1():
        """the held conv1 weight gradients as one launch (side stream when that is on)"""
        if not w1_held:
            return
        held = list(w1_held)
        del w1_held[:]
        with side:
            # one grid for all of them only where sep_pw_wgrad_batch has a batched kernel (the fp16 producer / consumer form: f16x3, 256 x 128 tiles);
            # otherwise the entry point issues one launch per product and every product needs the slabs of a launch of its own
            batched = (len(held) > 1 and hasattr(K, "pw_wgrad_batch") and sepkernels.gemm_arith() == sepkernels.ARITH_F16X3
                       and H % 256 == 0 and Bn % 128 == 0)
            ns = _nsplit(H, Bn, chunks, target_blocks=max(8, 512 // len(held)) if batched else 512)
            calls, segs = [], []
            for da_k, x_k, pre_k in held:
                part = torch.empty(ns, H, Bn, **f32)
                pb = torch.empty(ns, H, **f32)
                calls.append(dict(B=B, M=H, N=Bn, T=F, ldt=ldt, G=da_k, X=x_k, partial=part, partial_bias=pb, nsplit=ns, eps=eps))
                segs += [(part, 0, G[pre_k + "bottleneck_conv1d.weight"], H * Bn, ns, H * Bn, 0, 1.0),
                         (pb, 0, G[pre_k + "bottleneck_conv1d.bias"], H, ns, H, 0, 1.0)]
            if batched:
                K.pw_wgrad_batch(calls)
            else:
                for c in calls:
                    K.pw_wgrad(**c)
            if side.on:
                K.reduce_slabs(segs)
            else:
                pending.extend(segs)        # (main stream: one pool -- nothing to lend)
    deferred = []                  # leaves of the layer just differentiated, queued on the side stream behind the next layer's hand-off
    finals = []                    # queued sep_gln_bwd_finalize calls (flushed with `pending`, in front of it)
    flushed_from = nl + 1          # dalpha entries [flushed_from, nl] are already converted (bucketed mode)
    dout = None
    for li in range(nl - 1, -1, -1):
        pre, dil, dual = layers[li]
        sp = pre + "separable_conv1d."
        x, a, z = sv.acts[li]
        st1, st2 = stats[1 + 2 * li], stats[2 + 2 * li]
        g1, b1, al1 = P[pre + "norm1d.norm.weight"], P[pre + "norm1d.norm.bias"], P[pre + "nonlinear1d.weight"]
        g2, b2, al2 = P[sp + "norm1d.norm.weight"], P[sp + "norm1d.norm.bias"], P[sp + "nonlinear1d.weight"]
        Ws = P[sp + "skip_pointwise_conv1d.weight"]
        cnt = H * F

        # The heads' weight gradients first, taken against u2 = PReLU(z) (the gain and shift of gLN2 left out) on sample-aligned slabs:
        # their contractions ARE the row sums gLN2's backward needs (sep_gln_bwd_from_wgrad: R1 = W^T gs, R2 = sum_m W * raw), so the
        # input-gradient product below has no row-sum epilogue and never reads z, and no second-stage kernel runs for this gLN.
        pbeta2 = torch.empty(B, H, **f32)
        pgamma2 = torch.empty(B, H, **f32)
        xkw = dict(x_mode=PRO_PRELU, x_alpha=al2, weps=teps, aligned=True)
        heads = []      # (G, G2, g_split, rows, W as one [rows][H] matrix, [(gradient tensor, first row, rows)], [(bias gradient, first row, rows)])
        if dual:
            Wo = P[sp + "output_pointwise_conv1d.weight"]
            if Bn % 128 == 0 and _adjacent(Wo, Ws):
                heads.append((dout, dS, Bn, Bn + Sc, Wo.as_strided((Bn + Sc, H), (H, 1)),
                              [(sp + "output_pointwise_conv1d.weight", 0, Bn), (sp + "skip_pointwise_conv1d.weight", Bn, Sc)],
                              [(sp + "output_pointwise_conv1d.bias", 0, Bn), (sp + "skip_pointwise_conv1d.bias", Bn, Sc)]))
            else:
                heads.append((dout, None, 0, Bn, Wo, [(sp + "output_pointwise_conv1d.weight", 0, Bn)], [(sp + "output_pointwise_conv1d.bias", 0, Bn)]))
                heads.append((dS, None, 0, Sc, Ws, [(sp + "skip_pointwise_conv1d.weight", 0, Sc)], [(sp + "skip_pointwise_conv1d.bias", 0, Sc)]))
        else:
            heads.append((dS, None, 0, Sc, Ws, [(sp + "skip_pointwise_conv1d.weight", 0, Sc)], [(sp + "skip_pointwise_conv1d.bias", 0, Sc)]))
        # Side stream: [heads weight gradient -> gLN2 sums] run beside the input-gradient product below (both only need dout, dS, z); the
        # depthwise backward waits for the sums through an event.  The PREVIOUS layer's conv1 weight gradient -- a leaf nobody waits for --
        # is queued behind them, so that it never delays the hand-off.
        segs = []
        dWbs = [torch.empty(B, rows, H, **f32) for (_, _, _, rows, _, _, _) in heads]
        side.keep(pbeta2, pgamma2, *dWbs)
        with side:
            for hi, (Gt, G2t, gsp, rows, Wmat, wnames, bnames) in enumerate(heads):
                pre_kw = dict(G2_pre=dS_pre) if (dS_pre is not None and G2t is dS and gsp == Bn and rows == Bn + Sc) else {}
                part, pb, ns = wgrad(rows, H, Gt, z, True, True, G2=G2t, g_split=gsp, **pre_kw, **xkw)
                side.lend(pb)           # reduced on the MAIN stream by the flush of `pending`
                K.gln_bwd_from_wgrad(part, pb, Wmat, st2, g2, b2, cnt, teps, dWbs[hi], pbeta2, pgamma2, bacc[2 + 2 * li], arrive[2 + 2 * li],
                                     bsum[2 + 2 * li], B, rows, H, ns // B, accumulate=int(hi > 0), products=len(heads))
                segs += [(dWbs[hi], r0 * H, G[nm], nr * H, B, rows * H, 0, 1.0) for nm, r0, nr in wnames]
                segs += [(pb, r0, G[nm], nr, ns, rows, 0, 1.0) for nm, r0, nr in bnames]
        pending += segs
        ev_sums = side.mark()
        for leaf in deferred:
            leaf()
        deferred = []

        # dv2 = Wo^T dout + Ws^T dS
        dv2 = torch.empty(B, H, ldt, **f32)
        if dual:
            if "heads.{}^T".format(li) in PK:
                K.pw_gemm(B=B, M=H, K=Bn + Sc, T=F, ldt=ldt, trans_a=1, A=Wo, A2=Ws, A_pk=PK["heads.{}^T".format(li)], X=dout, X2=dS,
                          k_split=Bn, Y=dv2, eps=teps)
            else:
                K.pw_gemm(B=B, M=H, K=Bn + Sc, T=F, ldt=ldt, trans_a=1, A=Wo, A2=Ws, X=dout, X2=dS, k_split=Bn, Y=dv2, eps=teps)
        else:
            K.pw_gemm(B=B, M=H, K=Sc, T=F, ldt=ldt, trans_a=1, A=Ws, A_pk=PK.get("skip.{}^T".format(li)), X=dS, Y=dv2, eps=teps)

        # depthwise^T and everything hanging off it (needs gLN2's sums from the side stream)
        side.wait(ev_sums)
        dv1 = torch.empty(B, H, ldt, **f32)
        rp1 = torch.empty(B, H, nt1024, 8, **f32)
        K.dwconv_bwd(dv2, z, a, st1, g1, b1, al1, st2, g2, al2, bsum[2 + 2 * li], P[sp + "depthwise_conv1d.weight"], P[sp + "depthwise_conv1d.bias"], dv1, rp1,
                     bacc[1 + 2 * li], None, None, B, H, F, ldt, dil, teps)
        pbeta1 = torch.empty(B, H, **f32)
        pgamma1 = torch.empty(B, H, **f32)
        pextra = torch.empty(B * 4 * H + B + B * H, **f32)
        pending += [
            (pbeta2, 0, G[sp + "norm1d.norm.bias"], H, B, H, 0, 1.0),
            (pgamma2, 0, G[sp + "norm1d.norm.weight"], H, B, H, 0, 1.0),
            (pbeta1, 0, G[pre + "norm1d.norm.bias"], H, B, H, 0, 1.0),
            (pgamma1, 0, G[pre + "norm1d.norm.weight"], H, B, H, 0, 1.0),
            (pextra, 0, G[sp + "depthwise_conv1d.bias"], H, B, 4 * H, 0, 1.0),
            (pextra, H, G[sp + "depthwise_conv1d.weight"], 3 * H, B, 4 * H, 0, 1.0),
            (pextra, B * 4 * H, G[sp + "nonlinear1d.weight"], 1, B, 1, 0, 1.0),
        ]

        # dx = W1^T da (+ dout through the residual); da = gLN1/PReLU1 backward of dv1, formed in the GEMM prologue
        dx = torch.empty(B, Bn, ldt, **f32)
        # da overwrites dv1 in place when a single row tile covers all outputs (each X element is then read once);
        # with several row tiles the other tiles still need the untouched dv1, so da goes to its own buffer
        da = dv1 if Bn <= 128 else torch.empty_like(dv1)
        K.pw_gemm(B=B, M=Bn, K=H, T=F, ldt=ldt, trans_a=1, A=P[pre + "bottleneck_conv1d.weight"], A_pk=PK.get("conv1.{}^T".format(li)),
                  X=dv1, Y=dx, pro_mode=PRO_GLN_BWD, pro_stats=st1, pro_gamma=g1, pro_alpha=al1, pro_aux=a, pro_bacc=bacc[1 + 2 * li],
                  pro_store=da, pro_dalpha=dalpha[li:li + 1], count=cnt, eps=teps,
                  epi_flags=(EPI_RESIDUAL if dout is not None else 0), epi_res=dout)
        # da and dx now exist: the side stream may go on (this layer's dW1, the next layer's head gradients)
        side.fork()
        side.keep(da, dx, rp1, pbeta1, pgamma1, pextra)

        # second stage of gLN1's backward (parameter gradients only: a leaf): queued, all layers of a flush go out in one launch per stage
        finals.append((rp1, nt1024, 8, st1, g1, cnt, teps, None, pbeta1, pgamma1, pextra, B, H))

        # this layer's leaf: the conv1 weight gradient.  Nothing waits for it, so it is held back and issued together with its neighbours'
        # (sep_pw_wgrad_batch: W1_BATCH products in one launch, each with 1 / W1_BATCH of the slabs; da and x stay alive until then)
        w1_held.append((da, x, pre))
        if len(w1_held) >= W1_BATCH:
            deferred.append(flush_w1)
        dout = dx
        X_layers = cfg["sep_num_layers"]
        if on_ready is not None and li % X_layers == 0 and li > 0:
            # a whole TCN block is done: flush its queued reductions (and its PReLU slopes) so that the caller can
            # start the all-reduce of this bucket while the earlier blocks are still being differentiated
            for leaf in deferred:
                leaf()
            deferred = []
            flush_w1()
            side.join()
            blk = li // X_layers
            hi = nl + 1 if blk == cfg["sep_num_blocks"] - 1 else (blk + 1) * X_layers     # the last block also owns the mask PReLU slope
            lo = li
            d32 = torch.empty(hi - lo, **f32)
            K.f64_to_f32(dalpha[lo:hi], d32, hi - lo, 0)
            pending += [(d32, q - lo, G[layers[q][0] + "nonlinear1d.weight"], 1, 1, 1, 0, 1.0) for q in range(lo, min(hi, nl))]
            if hi == nl + 1:
                pending.append((d32, nl - lo, G["separator.prelu.weight"], 1, 1, 1, 0, 1.0))
            K.gln_bwd_finalize_batch(finals)
            finals = []
            K.reduce_slabs(pending)
            pending = []
            flushed_from = lo
            on_ready(blk)
    for leaf in deferred:
        leaf()
    flush_w1()
    side.join()
    # PReLU slope gradients were accumulated in fp64 (one scalar per layer + the mask PReLU): one conversion, then
    # scattered into the parameter gradients by the same flush
    rest = flushed_from if on_ready is not None else nl + 1
    dal32 = torch.empty(rest, **f32)
    K.f64_to_f32(dalpha[:rest], dal32, rest, 0)
    pending += [(dal32, li, G[layers[li][0] + "nonlinear1d.weight"], 1, 1, 1, 0, 1.0) for li in range(min(rest, nl))]
    if rest == nl + 1:
        pending.append((dal32, nl, G["separator.prelu.weight"], 1, 1, 1, 0, 1.0))
    if finals:
        K.gln_bwd_finalize_batch(finals)
    K.reduce_slabs(pending)

    # ---- head: bottleneck conv, first gLN, encoder ------------------------------------------------------
    return head_backward(cfg, P, geo, stats[0], w, mixture, dout, dwm, G, PK, want_dmix)
