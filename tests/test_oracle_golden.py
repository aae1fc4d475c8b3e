"""CPU: pin the oracle (oracle/convtasnet_oracle.py) against vectors produced by the
real reference (oracle/make_golden.py, run in the build container) and against the
reference's own deterministic self-test values (criterion/pit.py:226-375)."""
import os

import numpy as np
import pytest
import torch

from oracle import convtasnet_oracle as O
from oracle.make_golden import CONFIGS


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _params(blob, prefix="param/", dtype=torch.float64):
    return {k[len(prefix):]: torch.from_numpy(blob[k]).to(dtype) for k in blob.files if k.startswith(prefix)}


def test_ops_against_reference_modules(golden_dir):
    g = _load(golden_dir, "ops.npz")
    t = lambda k: torch.from_numpy(g[k])
    y = O.gln(t("gln_x"), t("gln_w"), t("gln_b"))
    assert torch.allclose(y, t("gln_y"), rtol=1e-10, atol=1e-10)
    p = _params(g, "tdcn/")
    y = O.tdcn(t("tdcn_x"), p, "", 2, 3)
    assert torch.allclose(y, t("tdcn_y"), rtol=1e-10, atol=1e-10)
    y = O.encoder(t("enc_x"), t("enc_w"), 3, relu=True)
    assert torch.allclose(y, t("enc_y"), rtol=1e-10, atol=1e-12)
    y = O.decoder(t("dec_x"), t("dec_w"), 3)
    assert torch.allclose(y, t("dec_y"), rtol=1e-10, atol=1e-12)
    y = O.neg_sisdr(t("sdr_x"), t("sdr_t"), batch_mean=False)
    assert torch.allclose(y, t("sdr_negsisdr"), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("name", ["tiny", "mid", "softmax"])
def test_model_forward_loss_grads(golden_dir, name):
    g = _load(golden_dir, "convtasnet_{}.npz".format(name))
    cfg = CONFIGS[name]
    p = _params(g)
    mixture, sources = torch.from_numpy(g["mixture"]), torch.from_numpy(g["sources"])
    out, loss, pattern, grads = O.train_step(p, cfg, mixture, sources, dtype=torch.float64)
    ref = torch.from_numpy(g["output_f64"])
    assert (out - ref).abs().max() <= 1e-9 * ref.abs().max()
    assert abs(loss.item() - float(g["loss_f64"])) < 1e-8
    assert np.array_equal(pattern.numpy(), g["pattern"])
    assert sum(v.numel() for v in p.values()) == int(g["num_parameters"])
    for k, gr in grads.items():
        r = torch.from_numpy(g["grad/" + k]).double()
        assert (gr - r).abs().max() <= 2e-6 * r.abs().max() + 1e-12, k   # fixture stored as f32
    # fp32 arm of the oracle vs the reference's shipped fp32 forward: 1e-3 relative is the north-star bar
    out32, _ = O.conv_tasnet(mixture, {k: v.float() for k, v in p.items()}, cfg)
    ref32 = torch.from_numpy(g["output_f32"])
    assert (out32 - ref32).abs().max() <= 1e-4 * ref32.abs().max()


def test_pit_known_answers(golden_dir):
    """criterion/pit.py self-test, seed 111: -4.6058 / 4.4252 / SinkPIT [11.1611, 10.4200, 10.5582, 9.9728]."""
    g = _load(golden_dir, "pit_kat.npz")
    x, t = torch.from_numpy(g["pit_x"]).double(), torch.from_numpy(g["pit_t"]).double()
    sis = lambda a, b, batch_mean=False: O.neg_sisdr(a, b, batch_mean=batch_mean, sign=+1.0)
    neg = lambda a, b, batch_mean=False: O.neg_sisdr(a, b, batch_mean=batch_mean)
    loss, pattern = O.pit(sis, x, t, maximize=True)
    assert abs(loss.item() - (-4.6058)) < 1e-4 and abs(loss.item() - float(g["pit_sisdr_loss"])) < 1e-5
    assert pattern.tolist() == [[1, 0], [1, 0], [0, 1], [0, 1]]
    x, t = torch.from_numpy(g["sink_x"]).double(), torch.from_numpy(g["sink_t"]).double()
    loss, pattern = O.pit(neg, x, t)
    assert abs(loss.item() - 4.4252) < 1e-4
    assert pattern.tolist() == [[1, 0, 2], [2, 1, 0], [0, 1, 2], [2, 1, 0]]
    loss, P = O.sinkpit(neg, x, t, coldness=1.0, iteration=10, batch_mean=False)
    assert np.allclose(loss.numpy(), [11.1611, 10.4200, 10.5582, 9.9728], atol=1e-4)
    assert P.argmax(2).tolist() == [[2, 0, 2], [0, 1, 0], [0, 1, 2], [2, 1, 0]]
    loss, P = O.sinkpit(sis, x, t, coldness=1.0, iteration=10, maximize=True, batch_mean=False)
    assert np.allclose(loss.numpy(), g["sinkpit_pos_loss"], atol=1e-5)


def test_sinkpit_gradient(golden_dir):
    g = _load(golden_dir, "pit_kat.npz")
    x = torch.from_numpy(g["sg_x"]).requires_grad_(True)
    t = torch.from_numpy(g["sg_t"])
    neg = lambda a, b, batch_mean=False: O.neg_sisdr(a, b, batch_mean=batch_mean)
    loss, P = O.sinkpit(neg, x, t, coldness=2.0, iteration=20)
    loss.backward()
    assert abs(loss.item() - float(g["sg_loss"])) < 1e-9
    assert np.allclose(x.grad.numpy(), g["sg_grad"], rtol=1e-8, atol=1e-12)
    assert np.array_equal(P.argmax(2).numpy(), g["sg_pattern"])


def test_roofline_constants():
    """paper-best 2-spk = 29.47 MFLOP/frame and 775,872 B/frame (fwd+bwd = 3x fwd)."""
    cfg = dict(n_basis=512, kernel_size=16, stride=8, sep_hidden_channels=512, sep_bottleneck_channels=128,
               sep_skip_channels=128, sep_kernel_size=3, sep_num_blocks=3, sep_num_layers=8, n_sources=2)
    assert 3 * O.flops_per_frame(cfg) == 29466624
    assert 3 * O.bytes_per_frame(cfg) == 775872
    assert O.num_frames(32000, 16, 8) == 3999


@pytest.mark.parametrize("name", ["tiny", "mid", "softmax"])
def test_fast_port_matches_reference_and_algebra_oracle(golden_dir, name):
    """oracle/fast_port.py (what bench.py times as cpu_baseline) against the reference golden vectors."""
    from oracle import fast_port as FP
    g = _load(golden_dir, "convtasnet_{}.npz".format(name))
    cfg = CONFIGS[name]
    p = _params(g)
    mixture, sources = torch.from_numpy(g["mixture"]), torch.from_numpy(g["sources"])
    out, loss, pattern, grads = FP.train_step(p, cfg, mixture, sources, dtype=torch.float64)
    ref = torch.from_numpy(g["output_f64"])
    assert (out - ref).abs().max() <= 1e-9 * ref.abs().max()
    assert abs(loss.item() - float(g["loss_f64"])) < 1e-8
    assert np.array_equal(pattern.numpy(), g["pattern"])
    for k, gr in grads.items():
        r = torch.from_numpy(g["grad/" + k]).double()
        assert (gr - r).abs().max() <= 2e-6 * r.abs().max() + 1e-12, k


def test_paper_b16_fixture_parameters_are_reproducible_here():
    """tests/golden/convtasnet_paper_b16.npz stores no parameters: the GPU test rebuilds them from the seeds with the PRODUCT's class.  Here
    (no GPU): that construction reproduces the reference's parameter fingerprints and input head, i.e. the fixture can be used at all."""
    import numpy as np
    from oracle.make_golden import PAPER_CFG, PAPER_B16_SEEDS, paper_b16_inputs
    from models.conv_tasnet import ConvTasNet
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convtasnet_paper_b16.npz"))
    torch.manual_seed(PAPER_B16_SEEDS["model"])
    model = ConvTasNet(**PAPER_CFG)
    mixture, sources = paper_b16_inputs(model)
    assert np.array_equal(mixture.numpy()[:, 0, :8], fx["mixture_head"])
    for k, v in model.state_dict().items():
        got = np.array([v.double().sum().item(), v.double().abs().sum().item()])
        assert np.allclose(got, fx["pfp/" + k], rtol=1e-9, atol=1e-12), k
    assert fx["pattern"].shape == (16, 2) and fx["per_utt_f64"].shape == (16,)
    assert len([k for k in fx.files if k.startswith("gfp/")]) == len(list(model.parameters()))
