"""CPU: criterion/distance.py and criterion/sdr.py:{sdr,SDR,NegSDR} through the CPU emulator of the C ABI, against the
formulas of reference src/criterion/distance.py:7-285 / src/criterion/sdr.py:6-120 written out with torch autograd
(value and gradient), and -- when the reference checkout is present in this container -- against the reference
classes themselves."""
import importlib.util
import os

import pytest
import torch

import sepkernels
from emulator import EmuBackend
from criterion.distance import L1Loss, L2Loss, MeanAbsoluteError, MeanSquaredError
from criterion.sdr import NegSDR, SDR, sdr
from criterion.pit import PIT1d


@pytest.fixture(autouse=True)
def emu():
    old = sepkernels._set_backend_for_tests(EmuBackend())
    yield
    sepkernels._set_backend_for_tests(old)


def _formula(name, x, t, dim):
    d = x - t
    if name == "mae":
        return d.abs().mean(dim=dim)
    if name == "mse":
        return (d ** 2).mean(dim=dim)
    if name == "l1":
        return d.abs().sum(dim=dim)
    if name == "l2":
        return torch.sqrt((d.abs() ** 2).sum(dim=dim))
    raise KeyError(name)


CLASSES = {"mae": MeanAbsoluteError, "mse": MeanSquaredError, "l1": L1Loss, "l2": L2Loss}


def _reduce(loss, reduction, batch_mean):
    if reduction and loss.dim() > 1:
        rest = tuple(range(1, loss.dim()))
        loss = loss.mean(dim=rest) if reduction == "mean" else loss.sum(dim=rest)
    return loss.mean(dim=0) if batch_mean else loss


@pytest.mark.parametrize("name", ["mae", "mse", "l1", "l2"])
@pytest.mark.parametrize("shape,dim", [((3, 2, 101), -1), ((2, 4, 2, 64), -1), ((3, 5, 40), 1), ((2, 3, 4, 10), (2, 3))])
@pytest.mark.parametrize("reduction,batch_mean", [("mean", True), ("sum", False)])
def test_distance_value_and_gradient(name, shape, dim, reduction, batch_mean):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(*shape, generator=g, dtype=torch.float64)
    t = torch.randn(*shape, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    want = _reduce(_formula(name, xr, t, dim), reduction, batch_mean)
    want.sum().backward()

    xs = x.clone().requires_grad_(True)
    crit = CLASSES[name](dim=dim, reduction=reduction)
    assert crit.maximize is False
    got = crit(xs, t, batch_mean=batch_mean)
    got.sum().backward()
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(xs.grad, xr.grad, rtol=2e-7, atol=1e-12)      # row coefficients cross the ABI as fp32


def test_mean_errors_without_reduction_keep_remaining_axes():
    x, t = torch.randn(2, 3, 50), torch.randn(2, 3, 50)
    out = MeanSquaredError(dim=-1)(x, t, batch_mean=False)
    assert out.shape == (2, 3)
    torch.testing.assert_close(out, ((x - t) ** 2).mean(-1))
    with pytest.raises(ValueError):
        L1Loss(reduction=None)
    with pytest.raises(NotImplementedError):
        MeanAbsoluteError(dim=-1)(x, t.requires_grad_(True))


@pytest.mark.parametrize("shape", [(4, 300), (3, 2, 257), (2, 3, 2, 64)])
def test_sdr_value_and_gradient(shape):
    g = torch.Generator().manual_seed(9)
    t = torch.randn(*shape, generator=g, dtype=torch.float64)
    x = t + 0.3 * torch.randn(*shape, generator=g, dtype=torch.float64)
    eps = 1e-12
    xr = x.clone().requires_grad_(True)
    want = 10 * torch.log10(((t ** 2).sum(-1) + eps) / (((t - xr) ** 2).sum(-1) + eps))
    want.sum().backward()
    xs = x.clone().requires_grad_(True)
    got = sdr(xs, t)
    got.sum().backward()
    torch.testing.assert_close(got, want.detach(), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(xs.grad, xr.grad, rtol=2e-7, atol=1e-12)
    # module reductions: over sources (and mics), then batch
    pos, neg = SDR(reduction="mean"), NegSDR(reduction="sum")
    assert pos.maximize is True and neg.maximize is False
    w = want.detach()
    rest = tuple(range(1, w.dim()))
    torch.testing.assert_close(pos(x, t), (w.mean(dim=rest) if rest else w).mean(0))
    torch.testing.assert_close(neg(x, t, batch_mean=False), -(w.sum(dim=rest) if rest else w))


def test_pit_over_negsdr_and_mse_uses_the_generic_path():
    g = torch.Generator().manual_seed(2)
    t = torch.randn(3, 2, 200, generator=g)
    x = (t[:, [1, 0]] + 0.1 * torch.randn(3, 2, 200, generator=g)).requires_grad_(True)
    loss, pattern = PIT1d(NegSDR(), n_sources=2)(x, t)
    assert pattern.tolist() == [[1, 0]] * 3
    loss.backward()
    assert torch.isfinite(x.grad).all()
    loss2, pattern2 = PIT1d(MeanSquaredError(dim=-1, reduction="mean"), n_sources=2)(x.detach(), t)
    assert pattern2.tolist() == [[1, 0]] * 3
    torch.testing.assert_close(loss2, ((x.detach() - t[:, [1, 0]]) ** 2).mean())


REF = "/root/reference/src/criterion"


def _load_reference(fname, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_against_reference_classes():
    rd, rs = _load_reference("distance.py", "_ref_distance"), _load_reference("sdr.py", "_ref_sdr")
    g = torch.Generator().manual_seed(77)
    t = torch.randn(3, 4, 2, 333, generator=g, dtype=torch.float64)      # the music recipe's (batch, sources, channels, T)
    x = t + 0.5 * torch.randn(3, 4, 2, 333, generator=g, dtype=torch.float64)
    pairs = [(MeanAbsoluteError(dim=-1, reduction="mean"), rd.MeanAbsoluteError(dim=-1, reduction="mean")),
             (MeanSquaredError(dim=-1, reduction="mean"), rd.MeanSquaredError(dim=-1, reduction="mean")),
             (MeanSquaredError(dim=(2, 3), reduction="sum"), rd.MeanSquaredError(dim=(2, 3), reduction="sum")),
             (L1Loss(dim=3), rd.L1Loss(dim=3)), (L2Loss(dim=1, reduction="sum"), rd.L2Loss(dim=1, reduction="sum")),
             (SDR(), rs.SDR()), (NegSDR(reduction="sum"), rs.NegSDR(reduction="sum"))]
    for mine, ref in pairs:
        for batch_mean in (True, False):
            a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            la, lb = mine(a, t, batch_mean=batch_mean), ref(b, t, batch_mean=batch_mean)
            assert la.shape == lb.shape and mine.maximize == ref.maximize
            torch.testing.assert_close(la, lb, rtol=1e-12, atol=1e-12)
            la.sum().backward(); lb.sum().backward()
            torch.testing.assert_close(a.grad, b.grad, rtol=2e-7, atol=1e-13)
