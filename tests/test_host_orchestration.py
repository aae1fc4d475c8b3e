"""CPU: the host-side launch sequence (sepkernels/net.py) driven through the CPU emulator of the C ABI
(tests/emulator.py) must reproduce the reference's forward, loss and every parameter gradient
(golden vectors generated from the real reference by oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import sepkernels
from sepkernels import net
from emulator import EmuBackend
from oracle import convtasnet_oracle as O
from oracle.make_golden import CONFIGS


@pytest.fixture()
def emu():
    old = sepkernels._set_backend_for_tests(EmuBackend())
    yield
    sepkernels._set_backend_for_tests(old)


def _run(golden_dir, name, dtype):
    g = np.load(os.path.join(golden_dir, "convtasnet_{}.npz".format(name)))
    cfg = dict(CONFIGS[name])
    net.check_supported(cfg)
    P = {k[6:]: torch.from_numpy(g[k]).to(dtype) for k in g.files if k.startswith("param/")}
    mixture = torch.from_numpy(g["mixture"]).to(dtype)
    sources = torch.from_numpy(g["sources"]).to(dtype)
    est, latent, sv = net.forward(cfg, P, mixture, want_latent=True, save=True)
    B, n_src = est.shape[0], est.shape[1]
    est3 = est.reshape(B, n_src, -1).detach().clone().requires_grad_(True)
    loss, pattern = O.pit(lambda a, b, batch_mean=False: O.neg_sisdr(a, b, batch_mean=batch_mean), est3, sources)
    loss.backward()
    G = {k: torch.full_like(v, float("nan")) for k, v in P.items()}
    net.backward(cfg, P, sv, est3.grad.reshape(est.shape), G)
    return g, est, latent, loss, pattern, G, sv


@pytest.mark.parametrize("name", ["tiny", "mid", "softmax"])
def test_forward_backward_fp64(golden_dir, emu, name):
    g, est, latent, loss, pattern, G, sv = _run(golden_dir, name, torch.float64)
    ref = torch.from_numpy(g["output_f64"])
    assert (est.reshape(ref.shape) - ref).abs().max() <= 1e-9 * ref.abs().max()
    F = sv.geo.F
    assert abs(latent[..., :F].sum().item() - float(g["latent_f64_sum"])) <= 1e-8 * float(g["latent_f64_abs_sum"])
    assert latent[..., F:].abs().max() == 0
    assert abs(loss.item() - float(g["loss_f64"])) < 1e-8
    assert np.array_equal(pattern.numpy(), g["pattern"])
    for k, v in G.items():
        r = torch.from_numpy(g["grad/" + k]).double()
        assert torch.isfinite(v).all(), k
        assert (v - r).abs().max() <= 2e-6 * r.abs().max() + 1e-12, k


def test_forward_backward_fp32_within_north_star_tolerance(golden_dir, emu):
    """Same sequence in fp32: 1e-3 relative on the forward (north_star), flat-vector 1e-3 on the gradients."""
    g, est, latent, loss, pattern, G, sv = _run(golden_dir, "mid", torch.float32)
    ref = torch.from_numpy(g["output_f64"])
    assert (est.reshape(ref.shape).double() - ref).abs().max() <= 1e-3 * ref.abs().max()
    num = max((G[k].double() - torch.from_numpy(g["grad/" + k]).double()).abs().max().item() for k in G)
    den = max(np.abs(g["grad/" + k]).max() for k in G)
    assert num <= 1e-3 * den


def test_pad_frames_are_zero(golden_dir, emu):
    g, est, latent, loss, pattern, G, sv = _run(golden_dir, "mid", torch.float64)
    F = sv.geo.F
    assert sv.geo.ldt % 128 == 0 and sv.geo.ldt >= F
    for t in [sv.w, sv.m, sv.skip] + [u for act in sv.acts for u in act]:
        assert t[..., F:].abs().max() == 0


def test_weight_bound_over_a_flat_buffer_and_over_scattered_tensors():
    """net.amax_over: the |A| bound of SEP_ARITH_F16X3 -- one reduction over the span when the tensors are views of one
    buffer (gaps included), per tensor otherwise; always >= every element's magnitude."""
    flat = torch.zeros(300)
    a, b, c = flat[16:116].view(10, 10), flat[128:160], flat[200:300].view(4, 25)
    torch.manual_seed(0)
    for t in (a, b, c):
        t.copy_(torch.randn(t.shape))
    b[3] = -7.5
    flat[180] = 99.0                                       # inside the span, in no tensor: may raise the bound, never lower it
    got = net.amax_over([c, a, b])
    assert got.shape == (1,) and got.item() == 99.0
    flat[180] = 0.0
    assert net.amax_over([c, a, b]).item() == 7.5
    scattered = [torch.randn(5, 5), torch.full((3,), -11.0), torch.randn(2, 2, 2)]
    assert net.amax_over(scattered).item() == 11.0
    far = torch.zeros(1_000_000)
    far[500_000] = 5.0
    assert net.amax_over([far[:4], far[-4:]]).item() == 0.0          # same buffer, but a span 125000x the data: per tensor
