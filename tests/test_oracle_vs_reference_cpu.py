"""CPU: the two oracle statements (oracle/convtasnet_oracle.py: elementary tensor algebra; oracle/fast_port.py: the same path on
torch.nn.functional) against the LIVE, unmodified reference at the PAPER-BEST configuration (N512 L16 B128 H512 Sc128 P3 X8 R3:
dilations up to 128, three repeats, the last layer without output head), forward, PIT loss, permutation and every parameter
gradient in fp64 -- the committed golden fixtures pin them only up to R=2, X=3.  Skipped where /root/reference is absent (the
GPU box); there the port is what the full-size GPU parity tests and bench.py's cpu_baseline rely on, which is why this
equality is pinned here."""
import os
import sys
import types

import pytest
import torch

from oracle import convtasnet_oracle as O
from oracle import fast_port as FP

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

PAPER = dict(n_basis=512, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
             sep_hidden_channels=512, sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3,
             sep_num_blocks=3, sep_num_layers=8, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
             sep_norm=True, mask_nonlinear="sigmoid", n_sources=2)


def _reference_classes():
    """The reference's ConvTasNet / criteria, imported from /root/reference/src under a private module namespace so that this
    repository's own `models`, `criterion`, ... packages (already imported by other tests) are left alone."""
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("models", "criterion", "modules", "utils", "transforms", "algorithm", "conv", "norm")}
    for k in saved:
        del sys.modules[k]
    stub = "torchaudio" not in sys.modules           # the reference imports it transitively without using it on this path
    if stub:
        sys.modules["torchaudio"] = types.ModuleType("torchaudio")
    path = list(sys.path)
    sys.path[:] = [REF] + [p for p in path if "dnn-based_source_separation_amd" not in p]
    try:
        from models.conv_tasnet import ConvTasNet
        from criterion.sdr import NegSISDR
        from criterion.pit import PIT1d
        assert sys.modules["models.conv_tasnet"].__file__.startswith(REF)
        return ConvTasNet, NegSISDR, PIT1d
    finally:
        sys.path[:] = path
        if stub:
            del sys.modules["torchaudio"]
        for k in [k for k in sys.modules if k.split(".")[0] in ("models", "criterion", "modules", "utils", "transforms", "algorithm", "conv", "norm")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_oracle_and_port_equal_the_live_reference_at_paper_best():
    ConvTasNet, NegSISDR, PIT1d = _reference_classes()
    torch.manual_seed(111)
    ref = ConvTasNet(**PAPER).double()
    with torch.no_grad():                                  # default gamma = 1, beta = 0, alpha = 0.25 would hide indexing mistakes
        g = torch.Generator().manual_seed(7)
        for name, p in ref.named_parameters():
            if name.endswith("norm.weight") or name.endswith("norm.bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g, dtype=torch.float64))
            elif name.endswith("nonlinear1d.weight") or name.endswith("prelu.weight"):
                p.add_(0.1 * torch.rand(p.shape, generator=g, dtype=torch.float64))
    T = 20003                                              # off the stride grid: the input-padding branch
    sources = 0.1 * torch.randn(1, 2, T, generator=torch.Generator().manual_seed(222), dtype=torch.float64)
    mixture = sources.sum(1, keepdim=True)
    out = ref(mixture)
    loss, pattern = PIT1d(NegSISDR(), n_sources=2)(out, sources)
    loss.backward()
    p = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    grads = {k: q.grad for k, q in ref.named_parameters()}

    for name, step in (("fast_port", FP.train_step), ("algebra oracle", O.train_step)):
        o, l, pat, gr = step(p, PAPER, mixture, sources, dtype=torch.float64)
        assert (o - out.detach()).abs().max() <= 1e-12 * out.detach().abs().max(), name
        assert abs(l.item() - loss.item()) <= 1e-12 * abs(loss.item()), name
        assert torch.equal(torch.as_tensor(pat), pattern), name
        for k, gref in grads.items():
            err = (gr[k] - gref).abs().max().item()
            assert err <= 1e-10 * max(gref.abs().max().item(), 1e-30), "{}: {} {:.2e}".format(name, k, err / gref.abs().max().item())
