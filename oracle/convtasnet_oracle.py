"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement of the reference's Conv-TasNet separation path
(tky823/DNN-based_source_separation v0.7.2, pure PyTorch) written from the
algebra, not from the reference's module tree: every op is spelled out with
elementary tensor arithmetic (sums, products, einsum, slicing) so that it is an
independent statement of WHAT the reference computes.  It runs in any float
dtype (tests use float64 as the ground truth and float32 as the "same
arithmetic as the reference" arm) and is differentiable through torch
autograd, which is how gradient parity of the HIP backward kernels is checked.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this file.  The product (`dnn-based_source_separation_amd/`) never
does.

Pinning: the oracle is validated against the *real* reference, imported from
/root/reference in the build container by `oracle/make_golden.py`, which
writes the input/output vectors under `tests/golden/`;
`tests/test_oracle_golden.py` re-checks the oracle against those vectors on
every run (they travel to the GPU box, /root/reference does not).  It is also
pinned to the only deterministic printed values the reference owns, the
`src/criterion/pit.py:226-375` self-test (seed 111), stored in
`tests/golden/pit_kat.npz`.

Reference lines each function follows (all under /root/reference/src):
  gln                  modules/norm.py:11-29  (nn.GroupNorm(1, C, eps))
  prelu                models/tdcn.py:90,161 ; models/conv_tasnet.py:340 (nn.PReLU(), 1 parameter)
  pointwise_conv       models/tdcn.py:86,173,175 ; models/conv_tasnet.py:335,341 (nn.Conv1d k=1)
  depthwise_conv       models/tdcn.py:120-132 (zero pad after the norm) + :157 (groups=C dilated conv)
  residual_block       models/tdcn.py:107-147 + :177-196
  tdcn                 models/tdcn.py:29-41, 65-75
  encoder / decoder    models/filterbank.py:205-251
  separator            models/conv_tasnet.py:359-378
  conv_tasnet          models/conv_tasnet.py:121-171
  sisdr / neg_sisdr    criterion/sdr.py:122-139, 187-231
  pit                  criterion/pit.py:9-44
  sinkpit              criterion/pit.py:163-213
"""
import itertools

import torch

EPS = 1e-12


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def gln(x, gamma, beta, eps=EPS):
    """Global layer norm: per SAMPLE statistics over (C, T), biased variance,
    per-channel affine.  x (B, C, T)."""
    B = x.shape[0]
    flat = x.reshape(B, -1)
    mean = flat.mean(dim=1).view(B, 1, 1)
    var = ((flat - mean.view(B, 1)) ** 2).mean(dim=1).view(B, 1, 1)
    xhat = (x - mean) / torch.sqrt(var + eps)
    return xhat * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def prelu(x, a):
    """Scalar-parameter PReLU (one `a` shared by all channels)."""
    return torch.where(x > 0, x, a.reshape(()) * x)


def pointwise_conv(x, weight, bias=None):
    """1x1 convolution.  x (B, Cin, T), weight (Cout, Cin, 1)."""
    y = torch.einsum("oi,bit->bot", weight[:, :, 0], x)
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y


def depthwise_conv(v, weight, bias, dilation, causal=False):
    """Depthwise dilated conv with the reference's zero padding applied to the
    *normalised* signal: padding = (P-1)*d, split d|d for P=3 non-causal
    (tdcn.py:120-132).  v (B, C, T), weight (C, 1, P)."""
    B, C, T = v.shape
    P = weight.shape[-1]
    padding = (P - 1) * dilation
    if causal:
        left, right = padding, 0
    else:
        left = padding // 2
        right = padding - left
    vp = torch.zeros(B, C, T + padding, dtype=v.dtype)
    vp[:, :, left:left + T] = v
    z = torch.zeros_like(v)
    for k in range(P):
        z = z + weight[:, 0, k].view(1, C, 1) * vp[:, :, k * dilation:k * dilation + T]
    return z + bias.view(1, C, 1)


def residual_block(x, p, prefix, dilation, dual_head, eps=EPS):
    """One TCN layer: 1x1 -> PReLU -> gLN -> (zero pad) depthwise -> PReLU ->
    gLN -> {output 1x1 + residual, skip 1x1}."""
    u1 = prelu(pointwise_conv(x, p[prefix + "bottleneck_conv1d.weight"], p[prefix + "bottleneck_conv1d.bias"]),
               p[prefix + "nonlinear1d.weight"])
    v1 = gln(u1, p[prefix + "norm1d.norm.weight"], p[prefix + "norm1d.norm.bias"], eps)
    s = prefix + "separable_conv1d."
    z = depthwise_conv(v1, p[s + "depthwise_conv1d.weight"], p[s + "depthwise_conv1d.bias"], dilation)
    u2 = prelu(z, p[s + "nonlinear1d.weight"])
    v2 = gln(u2, p[s + "norm1d.norm.weight"], p[s + "norm1d.norm.bias"], eps)
    skip = pointwise_conv(v2, p[s + "skip_pointwise_conv1d.weight"], p[s + "skip_pointwise_conv1d.bias"])
    if dual_head:
        out = pointwise_conv(v2, p[s + "output_pointwise_conv1d.weight"], p[s + "output_pointwise_conv1d.bias"]) + x
    else:
        out = None
    return out, skip


def tdcn(x, p, prefix, num_blocks, num_layers, eps=EPS):
    """R blocks x X layers, dilation 2**layer, every layer's skip summed; the
    very last layer has no output head."""
    skip_sum = None
    for r in range(num_blocks):
        for l in range(num_layers):
            last = (r == num_blocks - 1) and (l == num_layers - 1)
            x, skip = residual_block(x, p, "{}net.{}.net.{}.".format(prefix, r, l), 2 ** l, not last, eps)
            skip_sum = skip if skip_sum is None else skip_sum + skip
    return skip_sum


def encoder(x, basis, stride, relu=False):
    """w[b,n,f] = sum_{c,k} basis[n,c,k] x[b,c,stride*f+k]  (+ReLU)."""
    B, Cin, T = x.shape
    N, _, L = basis.shape
    F_ = (T - L) // stride + 1
    idx = (torch.arange(F_) * stride).view(F_, 1) + torch.arange(L).view(1, L)  # (F, L)
    frames = x[:, :, idx]                       # (B, Cin, F, L)
    w = torch.einsum("nck,bcfk->bnf", basis, frames)
    return torch.clamp(w, min=0) if relu else w


def decoder(w_hat, basis, stride):
    """Basis synthesis + overlap-add (ConvTranspose1d, no bias).
    w_hat (B', N, F), basis (N, Cout, L) -> (B', Cout, (F-1)*stride+L)."""
    Bp, N, F_ = w_hat.shape
    _, Cout, L = basis.shape
    frames = torch.einsum("bnf,nck->bcfk", w_hat, basis)       # (B', Cout, F, L)
    T = (F_ - 1) * stride + L
    out = torch.zeros(Bp, Cout, T, dtype=w_hat.dtype)
    idx = ((torch.arange(F_) * stride).view(F_, 1) + torch.arange(L).view(1, L)).reshape(-1)
    out.index_add_(2, idx, frames.reshape(Bp, Cout, F_ * L))
    return out


def separator(w, p, cfg):
    n_src, N = cfg["n_sources"], cfg["n_basis"]
    eps = cfg.get("eps", EPS)
    x = gln(w, p["separator.norm1d.norm.weight"], p["separator.norm1d.norm.bias"], eps)
    x = pointwise_conv(x, p["separator.bottleneck_conv1d.weight"], p["separator.bottleneck_conv1d.bias"])
    x = tdcn(x, p, "separator.tdcn.", cfg["sep_num_blocks"], cfg["sep_num_layers"], eps)
    x = prelu(x, p["separator.prelu.weight"])
    x = pointwise_conv(x, p["separator.mask_conv1d.weight"], p["separator.mask_conv1d.bias"])
    if cfg.get("mask_nonlinear", "sigmoid") == "sigmoid":
        x = 1.0 / (1.0 + torch.exp(-x))
    else:  # softmax over the n_src*N channel axis, exactly as the reference does (dim=1)
        x = torch.softmax(x, dim=1)
    B, _, F_ = x.shape
    return x.view(B, n_src, N, F_)


def conv_tasnet(x, p, cfg):
    """Full forward.  x (B, Cin, T) -> (output (B, n_src, T), latent (B, n_src, N, T'))
    [single-channel path of conv_tasnet.py:121-171; in_channels>1 keeps the
    reference's view(B*n_src, N, T') -> decoder -> view(B, n_src, -1)]."""
    L, S = cfg["kernel_size"], cfg["stride"]
    n_src, N = cfg["n_sources"], cfg["n_basis"]
    B, Cin, T = x.shape
    padding = (S - (T - L) % S) % S
    pl = padding // 2
    pr = padding - pl
    xp = torch.zeros(B, Cin, T + padding, dtype=x.dtype)
    xp[:, :, pl:pl + T] = x
    w = encoder(xp, p["encoder.conv1d.weight"], S, relu=(cfg.get("enc_nonlinear") == "relu"))
    mask = separator(w, p, cfg)
    latent = w.unsqueeze(1) * mask
    xh = decoder(latent.reshape(B * n_src, N, -1), p["decoder.conv_transpose1d.weight"], S)
    xh = xh.reshape(B, n_src, -1)
    out = xh[:, :, pl:xh.shape[-1] - pr] if padding > 0 else xh
    return out, latent


# ----------------------------------------------------------------------------
# criteria
# ----------------------------------------------------------------------------
def sisdr(x, t, eps=EPS):
    """SI-SDR over the last axis, exactly in the reference's (two-pass) form."""
    alpha = (x * t).sum(-1, keepdim=True) / ((t ** 2).sum(-1, keepdim=True) + eps)
    num = ((alpha * t) ** 2).sum(-1) + eps
    den = ((alpha * t - x) ** 2).sum(-1) + eps
    return 10.0 * torch.log10(num / den)


def neg_sisdr(x, t, batch_mean=True, reduction="mean", eps=EPS, sign=-1.0):
    loss = sign * sisdr(x, t, eps)
    nd = x.dim()
    if reduction:
        if nd == 3:
            loss = loss.mean(1) if reduction == "mean" else loss.sum(1)
        elif nd == 4:
            loss = loss.mean((1, 2)) if reduction == "mean" else loss.sum((1, 2))
    if batch_mean:
        loss = loss.mean(0)
    return loss


def pit(criterion, x, t, maximize=False, batch_mean=True):
    """Exhaustive permutation search.  criterion(x, t_perm, batch_mean=False) -> (B,)."""
    n = x.shape[1]
    patterns = torch.tensor(list(itertools.permutations(range(n))), dtype=torch.long)
    losses = torch.stack([criterion(x, t[:, pat], batch_mean=False) for pat in patterns], dim=1)
    loss, idx = (losses.max(1) if maximize else losses.min(1))
    if batch_mean:
        loss = loss.mean(0)
    return loss, patterns[idx]


def sinkpit(pair_criterion, x, t, coldness=1.0, iteration=10, maximize=False, batch_mean=True):
    """Sinkhorn PIT.  pair_criterion(x_i, t_j) is evaluated for all n*n pairs
    (2-D inputs -> no source reduction, criterion/pit.py:170-173)."""
    B, n = x.shape[0], x.shape[1]
    xi = x.unsqueeze(2).expand(-1, -1, n, -1).reshape(B * n * n, -1)
    tj = t.unsqueeze(1).expand(-1, n, -1, -1).reshape(B * n * n, -1)
    C = pair_criterion(xi, tj, batch_mean=False).view(B, n, n)
    if maximize:
        C = -C
    Z = -coldness * C
    for _ in range(iteration):
        Z = Z - torch.logsumexp(Z, dim=1, keepdim=True)
        Z = Z - torch.logsumexp(Z, dim=2, keepdim=True)
    P = torch.exp(Z)
    loss = ((C + Z / coldness) * P).sum((1, 2))
    if maximize:
        loss = -loss
    if batch_mean:
        loss = loss.mean(0)
    return loss, P


# ----------------------------------------------------------------------------
# convenience: one training step (forward + PIT(NegSI-SDR) + backward)
# ----------------------------------------------------------------------------
def train_step(p, cfg, mixture, sources, dtype=torch.float64):
    """Returns (output, loss, pattern, grads dict) computed in `dtype`."""
    pp = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in p.items()}
    out, _ = conv_tasnet(mixture.to(dtype), pp, cfg)
    loss, pattern = pit(lambda a, b, batch_mean=False: neg_sisdr(a, b, batch_mean=batch_mean),
                        out, sources.to(dtype))
    loss.backward()
    grads = {k: v.grad for k, v in pp.items()}
    return out.detach(), loss.detach(), pattern, grads


def num_frames(T, L, S):
    padding = (S - (T - L) % S) % S
    return (T + padding - L) // S + 1


def flops_per_frame(cfg):
    """Forward MAC/frame formula of SURVEY.md section 8(d); fwd+bwd = 3x, FLOP = 2 MAC."""
    N, L = cfg["n_basis"], cfg["kernel_size"]
    Bn, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    P, X, R, ns = cfg["sep_kernel_size"], cfg["sep_num_layers"], cfg["sep_num_blocks"], cfg["n_sources"]
    mac = N * L + N * Bn + (R * X - 1) * (2 * Bn * H + H * Sc + H * P) + (Bn * H + H * Sc + H * P) + Sc * ns * N + ns * N * L
    return 2 * mac


def bytes_per_frame(cfg):
    """Forward algorithmic HBM bytes/frame of SURVEY.md section 8(d) (fp32)."""
    N = cfg["n_basis"]
    Bn, H, Sc = cfg["sep_bottleneck_channels"], cfg["sep_hidden_channels"], cfg["sep_skip_channels"]
    X, R, ns, S = cfg["sep_num_layers"], cfg["sep_num_blocks"], cfg["n_sources"], cfg["stride"]
    floats = R * X * (2 * Bn + 4 * H + 2 * Sc) + (2 * N + Bn + 2 * ns * N + ns * S)
    return 4 * floats
