"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Second CPU statement of the same path, written against torch.nn.functional (conv1d / group_norm / prelu /
conv_transpose1d), i.e. the same ATen CPU kernels (oneDNN convolution, native GroupNorm) the reference's modules
dispatch to.  It exists because the algebra oracle (convtasnet_oracle.py) spells every op out with elementary
tensor arithmetic and is ~10x slower than the reference on a CPU; this port is what `bench.py` times as
`cpu_baseline` (kind "port") so that the CPU number is representative of the reference's own CPU path, and what
the full-size GPU parity test uses as its fp64 ground truth.  It is pinned to the reference golden vectors and
to the algebra oracle by tests/test_oracle_golden.py.

Reference lines followed: models/conv_tasnet.py:121-171,359-378 ; models/tdcn.py:29-41,65-75,107-147,177-196 ;
modules/norm.py:18,27 ; models/filterbank.py:222-230,245-247 ; criterion/sdr.py:122-139,198-227 ; criterion/pit.py:9-44.
"""
import itertools

import torch
import torch.nn.functional as F

EPS = 1e-12


def _layer(x, p, pre, dilation, dual, eps):
    sp = pre + "separable_conv1d."
    T = x.shape[-1]
    h = F.conv1d(x, p[pre + "bottleneck_conv1d.weight"], p[pre + "bottleneck_conv1d.bias"])
    h = F.prelu(h, p[pre + "nonlinear1d.weight"])
    h = F.group_norm(h, 1, p[pre + "norm1d.norm.weight"], p[pre + "norm1d.norm.bias"], eps)
    P = p[sp + "depthwise_conv1d.weight"].shape[-1]
    pad = (P - 1) * dilation
    h = F.pad(h, (pad // 2, pad - pad // 2))
    h = F.conv1d(h, p[sp + "depthwise_conv1d.weight"], p[sp + "depthwise_conv1d.bias"], dilation=dilation, groups=h.shape[1])
    h = F.prelu(h, p[sp + "nonlinear1d.weight"])
    h = F.group_norm(h, 1, p[sp + "norm1d.norm.weight"], p[sp + "norm1d.norm.bias"], eps)
    skip = F.conv1d(h, p[sp + "skip_pointwise_conv1d.weight"], p[sp + "skip_pointwise_conv1d.bias"])
    out = None
    if dual:
        out = F.conv1d(h, p[sp + "output_pointwise_conv1d.weight"], p[sp + "output_pointwise_conv1d.bias"]) + x
    return out, skip


def conv_tasnet(x, p, cfg):
    L, S, n_src, N = cfg["kernel_size"], cfg["stride"], cfg["n_sources"], cfg["n_basis"]
    eps = cfg.get("eps", EPS)
    B, Cin, T = x.shape
    padding = (S - (T - L) % S) % S
    pl, pr = padding // 2, padding - padding // 2
    w = F.conv1d(F.pad(x, (pl, pr)), p["encoder.conv1d.weight"], stride=S)
    if cfg.get("enc_nonlinear") == "relu":
        w = F.relu(w)
    h = F.group_norm(w, 1, p["separator.norm1d.norm.weight"], p["separator.norm1d.norm.bias"], eps)
    h = F.conv1d(h, p["separator.bottleneck_conv1d.weight"], p["separator.bottleneck_conv1d.bias"])
    R, X = cfg["sep_num_blocks"], cfg["sep_num_layers"]
    skip_sum = 0
    for r in range(R):
        for l in range(X):
            last = r == R - 1 and l == X - 1
            # NOT cfg["eps"]: the reference's Separator builds its TimeDilatedConvNet without handing `eps` down (conv_tasnet.py:336-339), so
            # every gLN inside the TCN runs with tdcn.py's own default EPS whatever ConvTasNet(eps=...) says; "tcn_eps" is the key
            # sepkernels/net.py uses for the same fact
            h, skip = _layer(h, p, "separator.tdcn.net.{}.net.{}.".format(r, l), 2 ** l, not last, cfg.get("tcn_eps", EPS))
            skip_sum = skip_sum + skip
    h = F.prelu(skip_sum, p["separator.prelu.weight"])
    mask = F.conv1d(h, p["separator.mask_conv1d.weight"], p["separator.mask_conv1d.bias"])
    # reference conv_tasnet.py:353-357: Sigmoid, or Softmax(dim=1) over all n_src*N channels of a frame
    mask = torch.softmax(mask, dim=1) if cfg.get("mask_nonlinear", "sigmoid") == "softmax" else torch.sigmoid(mask)
    latent = w.unsqueeze(1) * mask.view(B, n_src, N, -1)
    xh = F.conv_transpose1d(latent.view(B * n_src, N, -1), p["decoder.conv_transpose1d.weight"], stride=S)
    xh = F.pad(xh.view(B, n_src, Cin, -1), (-pl, -pr))        # crop per channel, then the reference's views
    return (xh.view(B, n_src, -1) if Cin == 1 else xh), latent


def neg_sisdr_pit(est, src, eps=EPS):
    """PIT1d(NegSISDR()) : (loss (), pattern (B, n))."""
    n = est.shape[1]
    pats = torch.tensor(list(itertools.permutations(range(n))), dtype=torch.long)
    losses = []
    for pat in pats:
        t = src[:, pat]
        alpha = (est * t).sum(-1, keepdim=True) / ((t ** 2).sum(-1, keepdim=True) + eps)
        v = (((alpha * t) ** 2).sum(-1) + eps) / (((alpha * t - est) ** 2).sum(-1) + eps)
        losses.append((-10 * torch.log10(v)).mean(1))
    losses = torch.stack(losses, 1)
    loss, idx = losses.min(1)
    return loss.mean(0), pats[idx.cpu()]


def train_step(p, cfg, mixture, sources, dtype=torch.float32):
    pp = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in p.items()}
    out, _ = conv_tasnet(mixture.to(dtype), pp, cfg)
    loss, pattern = neg_sisdr_pit(out, sources.to(dtype))
    loss.backward()
    return out.detach(), loss.detach(), pattern, {k: v.grad for k, v in pp.items()}


def forward_loss(p, cfg, mixture, sources, dtype=torch.float32):
    """Forward + PIT(NegSI-SDR) only (no autograd tape): (output, loss, permutation) -- for full-batch forward comparisons where the
    gradients are judged elsewhere (tests/test_gpu_model.py::test_batch16_each_utterance_against_the_oracle)."""
    with torch.no_grad():
        q = {k: v.to(dtype) for k, v in p.items()}
        out, _ = conv_tasnet(mixture.to(dtype), q, cfg)
        loss, pattern = neg_sisdr_pit(out, sources.to(dtype))
    return out, loss, pattern

