"""CPU: configurations outside the fused kernel family -- causal (cLN: the reference constructor's default), non-separable
convolutions with P = 5, non-dilated / no norm / no activation -- run as the module-by-module composition (SURVEY.md section
8b fallback) and reproduce the reference: forward, PIT loss, permutation and every parameter gradient against golden vectors
generated from the unmodified reference (oracle/make_golden.py), in fp64 through the CPU emulator of the C ABI (gLN modules)
and plain torch operations (cLN, convolutions).  Also: stand-alone forwards of the TCN sub-modules, the decision of the
execution path at construction time, and nn.DataParallel replicas of the fused model."""
import os

import numpy as np
import pytest
import torch

import sepkernels
from emulator import EmuBackend
from oracle.make_golden import CONFIGS, COMPOSED, STAGED, DERIVED
from models.conv_tasnet import ConvTasNet
from models.tdcn import TimeDilatedConvNet, ResidualBlock1d
from modules.norm import CumulativeLayerNorm1d
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d


@pytest.fixture()
def emu():
    old = sepkernels._set_backend_for_tests(EmuBackend())
    yield
    sepkernels._set_backend_for_tests(old)


def _load(golden_dir, name, dtype=torch.float64):
    g = np.load(os.path.join(golden_dir, "convtasnet_{}.npz".format(name)))
    model = ConvTasNet(**CONFIGS[name])
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    return g, model.to(dtype)


@pytest.mark.parametrize("name", COMPOSED)
def test_composed_path_matches_the_reference(golden_dir, name, emu):
    g, model = _load(golden_dir, name)
    assert not model.fused and model.fused_reason          # decided at construction, with the reason on record
    assert model.fused_derived == (name in DERIVED)        # linear filterbanks run the fused sequence on derived bases
    assert model.num_parameters == int(g["num_parameters"])
    mixture, sources = torch.from_numpy(g["mixture"]).double(), torch.from_numpy(g["sources"]).double()
    est, latent = model.extract_latent(mixture)
    ref = torch.from_numpy(g["output_f64"])
    assert (est - ref).abs().max() <= 1e-9 * ref.abs().max()
    lsum = latent.sum().real.item() if torch.is_complex(latent) else latent.sum().item()      # (complex latent of the Fourier basis: real part of the sum)
    assert abs(lsum - float(g["latent_f64_sum"])) <= 1e-8 * float(g["latent_f64_abs_sum"])
    loss, pattern = PIT1d(NegSISDR(), n_sources=CONFIGS[name]["n_sources"])(est, sources)
    assert abs(loss.item() - float(g["loss_f64"])) <= 1e-9 * abs(float(g["loss_f64"]))
    assert np.array_equal(pattern.numpy(), g["pattern"])
    loss.backward()
    seen = 0
    for k, p in model.named_parameters():
        if "grad/" + k not in g.files:             # non-trainable parameters of the Fourier bases (time_seq; frequency when fixed)
            assert p.grad is None, k
            continue
        gr = torch.from_numpy(g["grad/" + k]).double()
        assert (p.grad - gr).abs().max() <= 2e-6 * max(gr.abs().max().item(), 1e-6), k          # the fixture stores fp64 gradients as fp32
        seen += 1
    assert seen == sum(1 for f in g.files if f.startswith("grad/"))


@pytest.mark.parametrize("name", STAGED)
def test_staged_causal_path_matches_the_reference(golden_dir, name, emu):
    """The causal family (the reference constructor's default) on channel counts the kernels take: runs layer by layer on the C ABI
    (models/conv_tasnet.py::_run_staged -- 1x1 convolutions on sep_pw_gemm, PReLU + cLN in sep_cln_*, the left-padded dilated taps in
    sep_depthwise_*), here through the CPU emulator in fp64, against golden vectors of the unmodified reference."""
    g, model = _load(golden_dir, name)
    assert not model.fused and model.staged and model.staged_reason is None
    mixture, sources = torch.from_numpy(g["mixture"]).double(), torch.from_numpy(g["sources"]).double()
    calls = []
    backend = sepkernels.backend()
    for fn in ("cln_fwd", "depthwise_fwd", "pw_gemm"):
        orig = getattr(backend, fn)
        setattr(backend, fn, (lambda o, n: (lambda *a, **k: (calls.append(n), o(*a, **k))[1]))(orig, fn))
    est, latent = model.extract_latent(mixture)
    nl = CONFIGS[name]["sep_num_blocks"] * CONFIGS[name]["sep_num_layers"]
    assert calls.count("cln_fwd") == 2 * nl + 1 and calls.count("depthwise_fwd") == nl          # the staged path ran, not the torch composition
    if name == "causal16_joint":          # 128-row bottleneck, adjacent weights: ONE product per layer for both heads (+ conv1; + bottleneck, mask)
        assert calls.count("pw_gemm") == 2 * nl + 2, calls.count("pw_gemm")
    ref = torch.from_numpy(g["output_f64"])
    assert (est - ref).abs().max() <= 1e-9 * ref.abs().max()
    assert abs(latent.sum().item() - float(g["latent_f64_sum"])) <= 1e-8 * float(g["latent_f64_abs_sum"])
    loss, pattern = PIT1d(NegSISDR(), n_sources=CONFIGS[name]["n_sources"])(est, sources)
    assert abs(loss.item() - float(g["loss_f64"])) <= 1e-9 * abs(float(g["loss_f64"]))
    assert np.array_equal(pattern.numpy(), g["pattern"])
    loss.backward()
    for k, p in model.named_parameters():
        gr = torch.from_numpy(g["grad/" + k]).double()
        assert (p.grad - gr).abs().max() <= 2e-6 * max(gr.abs().max().item(), 1e-6), k          # the fixture stores fp64 gradients as fp32


def test_cumulative_layer_norm_formula():
    """cLN against its definition written as an explicit loop over frames (reference modules/norm.py:58-96)."""
    torch.manual_seed(0)
    m = CumulativeLayerNorm1d(5, eps=1e-12).double()
    with torch.no_grad():
        m.gamma.copy_(torch.randn(1, 5, 1))
        m.beta.copy_(torch.randn(1, 5, 1))
    x = torch.randn(2, 5, 7, dtype=torch.float64)
    y = m(x)
    for t in range(7):
        seen = x[:, :, :t + 1]
        mu = seen.mean(dim=(1, 2))
        var = (seen ** 2).mean(dim=(1, 2)) - mu ** 2
        want = (x[:, :, t] - mu[:, None]) / (var.sqrt()[:, None] + 1e-12) * m.gamma[0, :, 0] + m.beta[0, :, 0]
        assert torch.allclose(y[:, :, t], want, rtol=1e-10, atol=1e-12)
    assert m(x.view(2, 5, 1, 7)).shape == (2, 5, 1, 7)
    with pytest.raises(ValueError):
        m(x[0])


def test_tcn_submodules_run_on_their_own(emu):
    """reference smoke tests tdcn.py:198-217: shapes of the stand-alone TCN and of one residual block (causal and not)."""
    x = torch.randn(2, 16, 50)
    for causal in (True, False):
        net = TimeDilatedConvNet(16, hidden_channels=24, skip_channels=12, kernel_size=3, num_blocks=2, num_layers=3, dilated=True,
                                 separable=True, causal=causal, nonlinear="prelu", norm=True)
        assert net(x).shape == (2, 12, 50)
        blk = ResidualBlock1d(16, hidden_channels=24, skip_channels=12, kernel_size=3, stride=1, dilation=4, separable=False,
                              causal=causal, nonlinear="prelu", norm=True, dual_head=True)
        out, skip = blk(x)
        assert out.shape == (2, 16, 50) and skip.shape == (2, 12, 50)
    # causality: the output up to frame t does not depend on later frames
    net = TimeDilatedConvNet(16, hidden_channels=24, skip_channels=12, num_blocks=1, num_layers=3, separable=True, causal=True,
                             nonlinear="prelu", norm=True).double()
    a = torch.randn(1, 16, 40, dtype=torch.float64)
    b = a.clone()
    b[..., 25:] += 1.0
    assert torch.allclose(net(a)[..., :25], net(b)[..., :25], atol=1e-12)


def test_reference_defaults_construct_and_run(emu):
    """ConvTasNet(512, 16) with the reference's defaults (causal=True) used to fail at the first forward: it now runs as the
    composition; the paper's non-causal configuration takes the fused path (decided in the constructor)."""
    m = ConvTasNet(32, 16, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None, sep_hidden_channels=32, sep_bottleneck_channels=16,
                   sep_skip_channels=16, sep_num_blocks=1, sep_num_layers=2)
    assert m.causal and not m.fused and "causal" in m.fused_reason
    assert m(torch.randn(1, 1, 800)).shape == (1, 2, 800)
    assert ConvTasNet(**CONFIGS["tiny"]).fused


def test_data_parallel_replica_finds_its_parameters(golden_dir, emu):
    """nn.DataParallel's replicate() leaves replica modules without registered parameters (the broadcast copies hang on them as
    plain attributes / `_former_parameters`): the fused forward must run on those, with gradients reaching the master copy.
    replicate() itself needs several GPUs; this builds the replica the way torch/nn/parallel/replicate.py does."""
    g = np.load(os.path.join(golden_dir, "convtasnet_tiny.npz"))
    model = ConvTasNet(**CONFIGS["tiny"])
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")})
    model = model.double()
    modules = list(model.modules())
    index = {m: i for i, m in enumerate(modules)}
    copies = [m._replicate_for_data_parallel() for m in modules]
    for m, c in zip(modules, copies):
        c._former_parameters = {}                         # replicate.py:157
        for key, child in m._modules.items():
            c._modules[key] = copies[index[child]]
        for key, p in m._parameters.items():
            if p is None:
                c._parameters[key] = None
                continue
            q = p * 1.0                                   # stands for Broadcast.apply: a non-leaf copy with a gradient edge to p
            setattr(c, key, q)
            c._former_parameters[key] = q
    replica = copies[0]
    assert not list(replica.parameters())
    mixture = torch.from_numpy(g["mixture"]).double()
    est = replica(mixture)
    ref = torch.from_numpy(g["output_f64"])
    assert (est - ref).abs().max() <= 1e-9 * ref.abs().max()
    sources = torch.from_numpy(g["sources"]).double()
    loss, _ = PIT1d(NegSISDR(), n_sources=2)(est, sources)
    loss.backward()
    for k, p in model.named_parameters():
        gr = torch.from_numpy(g["grad/" + k]).double()
        assert p.grad is not None and (p.grad - gr).abs().max() <= 2e-6 * max(gr.abs().max().item(), 1e-6), k


def test_cln_kernel_contract_via_emulator():
    """CumulativeLayerNorm1d through the C-ABI contract (sep_cln_fwd / sep_cln_bwd as restated by the CPU emulator): forward and
    all three gradients against autograd of the float64 composition (reference src/modules/norm.py:58-101)."""
    import sepkernels
    from emulator import EmuBackend
    from modules.norm import CumulativeLayerNorm1d
    old = sepkernels._set_backend_for_tests(EmuBackend())
    try:
        torch.manual_seed(5)
        B, C, T = 3, 24, 203
        m32 = CumulativeLayerNorm1d(C)
        with torch.no_grad():
            m32.gamma.copy_(torch.randn(1, C, 1))
            m32.beta.copy_(torch.randn(1, C, 1))
        m64 = CumulativeLayerNorm1d(C).double()
        m64.load_state_dict({k: v.double() for k, v in m32.state_dict().items()})
        x = (torch.randn(B, C, T) * torch.linspace(0.2, 3.0, T) + 0.7).requires_grad_(True)
        x64 = x.detach().double().requires_grad_(True)
        w = torch.randn(B, C, T)
        (m32(x) * w).sum().backward()
        (m64(x64) * w.double()).sum().backward()
        assert (m32(x).detach().double() - m64(x64).detach()).abs().max() < 2e-5
        for a, b in ((x.grad, x64.grad), (m32.gamma.grad, m64.gamma.grad), (m32.beta.grad, m64.beta.grad)):
            assert (a.double() - b).abs().max() <= 3e-5 * b.abs().max()
        # 4-D input (batch, C, S, chunk): same arithmetic on the flattened frames
        x4 = torch.randn(2, C, 7, 12)
        assert (m32(x4).double() - m64(x4.double())).abs().max() < 2e-5
    finally:
        sepkernels._set_backend_for_tests(old)


def test_gated_encoder_on_the_encoder_kernels_equals_the_convolutions(emu):
    """models.filterbank.GatedEncoder (reference src/models/filterbank.py:325-346): relu(U x) * sigmoid(V x) with both analysis convolutions on
    sep_encoder_fwd / sep_unfold + sep_pw_wgrad (sepkernels.functional.EncodeFn; here the emulator of the C ABI in fp64) against the same module
    on nn.Conv1d: output and both basis gradients; an input that is not whole frames, or that needs a gradient, keeps the convolutions."""
    from models.filterbank import GatedEncoder
    torch.manual_seed(4)
    enc = GatedEncoder(1, 32, kernel_size=16, stride=8).double()
    x = torch.randn(3, 1, 16 + 8 * 40, dtype=torch.float64)
    assert enc._on_kernels(x) and not enc._on_kernels(x[..., :-3]) and not enc._on_kernels(x.clone().requires_grad_(True))
    y = enc(x)
    w = torch.randn_like(y)
    gU, gV = torch.autograd.grad((y * w).sum(), [enc.conv1d_U.weight, enc.conv1d_V.weight])
    xn = x / (torch.linalg.norm(x, dim=2, keepdim=True) + enc.eps)
    ref = torch.relu(enc.conv1d_U(xn)) * torch.sigmoid(enc.conv1d_V(xn))
    rU, rV = torch.autograd.grad((ref * w).sum(), [enc.conv1d_U.weight, enc.conv1d_V.weight])
    assert y.shape == ref.shape == (3, 32, 41)
    assert (y - ref).abs().max() <= 1e-12 * ref.abs().max()
    assert (gU - rU).abs().max() <= 1e-10 * rU.abs().max() and (gV - rV).abs().max() <= 1e-10 * rV.abs().max()
    xs = x[..., :-3]
    xsn = xs / (torch.linalg.norm(xs, dim=2, keepdim=True) + enc.eps)
    assert torch.equal(enc(xs), torch.relu(enc.conv1d_U(xsn)) * torch.sigmoid(enc.conv1d_V(xsn)))
