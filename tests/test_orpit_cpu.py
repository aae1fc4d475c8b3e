"""CPU: the one-and-rest recursion (recipes/orpit.py) through the public module API on the C-ABI emulator, against the LIVE
reference model run through the reference driver's own loop (egs/wsj0-mix/orpit_conv-tasnet/src/adhoc_driver.py:193-207),
and the ORPIT training criterion against the reference's per-item python loop (src/criterion/pit.py:87-161)."""
import os

import pytest
import torch

import sepkernels
from emulator import EmuBackend
from oracle.make_golden import CONFIGS
from models.conv_tasnet import ConvTasNet
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d, ORPIT
from recipes.orpit import ORPITEvaluator, separate_one_and_rest
from test_oracle_vs_reference_cpu import REF, _reference_classes

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


@pytest.fixture()
def emu():
    old = sepkernels._set_backend_for_tests(EmuBackend())
    yield
    sepkernels._set_backend_for_tests(old)


def test_recursive_separation_equals_the_reference_loop(emu):
    RefNet, RefNegSISDR, RefPIT1d = _reference_classes()
    cfg = dict(CONFIGS["tiny"], n_sources=2)
    torch.manual_seed(3)
    ref = RefNet(**cfg).eval()
    mine = ConvTasNet(**cfg).eval()
    mine.load_state_dict(ref.state_dict())
    n_sources, B, T = 4, 2, 4000
    sources = 0.1 * torch.randn(B, n_sources, T)
    mixture = sources.sum(1, keepdim=True)
    with torch.no_grad():                                   # the reference Tester's loop, verbatim structure
        out = ref(mixture)
        one, rest = torch.split(out, [1, 1], dim=1)
        outs = [one]
        for _ in range(1, n_sources - 1):
            out = ref(rest)
            one, rest = torch.split(out, [1, 1], dim=1)
            outs.append(one)
        outs.append(rest)
        ref_out = torch.cat(outs, dim=1)
        ref_loss, ref_perm = RefPIT1d(RefNegSISDR(), n_sources=n_sources)(ref_out, sources, batch_mean=False)
        ref_mix, _ = RefPIT1d(RefNegSISDR(), n_sources=n_sources)(mixture, sources, batch_mean=False)
    res = ORPITEvaluator(mine, PIT1d(NegSISDR(), n_sources=n_sources), n_sources)(mixture, sources)
    assert res["output"].shape == (B, n_sources, T)
    assert (res["output"] - ref_out).abs().max() <= 1e-4 * ref_out.abs().max()      # three chained forwards
    assert torch.allclose(res["loss"], ref_loss, rtol=1e-4, atol=1e-4)
    assert torch.equal(res["perm_idx"], ref_perm)
    assert torch.allclose(res["loss_improvement"], ref_mix - ref_loss, rtol=1e-3, atol=1e-3)
    with pytest.raises(ValueError):
        separate_one_and_rest(ConvTasNet(**dict(cfg, n_sources=3)).eval(), mixture, 3)


def test_orpit_criterion_equals_the_reference_loop(emu):
    RefNet, RefNegSISDR, RefPIT1d = _reference_classes()
    torch.manual_seed(9)
    B, n, T = 3, 4, 2000
    est = torch.randn(B, 2, T, requires_grad=True)
    tgt = torch.randn(B, n, T)
    loss, idx = ORPIT(NegSISDR())(est, tgt, batch_mean=False)
    # restatement of pit.py:112-161 for equal n_sources: candidate idx = one vs target idx, rest vs the sum of the others
    crit = RefNegSISDR()
    cands = []
    for k in range(n):
        others = tgt.sum(1) - tgt[:, k]
        cands.append(crit(est[:, 0], tgt[:, k], batch_mean=False) + crit(est[:, 1], others, batch_mean=False) / (n - 1))
    cands = torch.stack(cands, 1)
    best, where = cands.min(1)
    assert torch.allclose(loss, best, rtol=1e-4, atol=1e-5) and torch.equal(idx.view(-1), where)
    loss.sum().backward()
    g = est.grad.clone()
    est.grad = None
    best.sum().backward()
    assert (g - est.grad).abs().max() <= 1e-4 * est.grad.abs().max()
