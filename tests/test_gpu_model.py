"""GPU: the drop-in ConvTasNet / criterion API running on libsepkernels.so against (a) the committed golden vectors
produced by the real reference, (b) the oracle on the same seeded inputs at the paper-best configuration, and
(c) size-independent properties at BASELINE.json's full size (B=16, 4 s @ 8 kHz).
Tolerance: 1e-3 relative (north_star: "within 1e-3 relative fp32") on the forward; gradients are judged on the
flat-vector rel-inf norm against an fp64 reference run (SURVEY.md section 8c noise-floor note)."""
import os

import numpy as np
import pytest
import torch

import sepkernels
from oracle.make_golden import CONFIGS, COMPOSED, STAGED, DERIVED
from oracle import fast_port as FP
from models.conv_tasnet import ConvTasNet
from criterion.sdr import NegSISDR, SISDR
from criterion.pit import PIT1d, SinkPIT
from modules.norm import GlobalLayerNorm

pytestmark = pytest.mark.gpu
TOL = 1e-3

PAPER = dict(n_basis=512, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
             sep_hidden_channels=512, sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3,
             sep_num_blocks=3, sep_num_layers=8, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
             sep_norm=True, mask_nonlinear="sigmoid", n_sources=2)


def _rel(a, b):
    return (a.double().cpu() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-30)


def _grad_report(model, ref_grads):
    num = den = 0.0
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        if k not in ref_grads:                 # non-trainable parameters (Fourier bases)
            continue
        r = ref_grads[k].double()
        e = (p.grad.double().cpu() - r).abs().max().item()
        num, den = max(num, e), max(den, r.abs().max().item())
        rel = e / (r.abs().max().item() + 1e-30)
        if rel > worst[1]:
            worst = (k, rel)
    return num / den, worst


def _oracle_fp32_noise(cfg, g):
    """Per-tensor |fp32 - fp64| of the ORACLE's own gradients (CPU port, same parameters and input): the reference's noise floor
    (up to 2.6e-3 on a scalar PReLU slope), which sets the per-tensor gate max(1e-3, 2 x this)."""
    p = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    _, _, _, g32 = FP.train_step(p, cfg, torch.from_numpy(g["mixture"]), torch.from_numpy(g["sources"]), dtype=torch.float32)
    out = {}
    for k, v in g32.items():
        r = torch.from_numpy(g["grad/" + k]).double()
        out[k] = (v.double() - r).abs().max().item() / (r.abs().max().item() + 1e-30)
    return out


@pytest.mark.parametrize("name,arith", [("tiny", "f16x3"), ("mid", "f16x3"), ("softmax", "f16x3"), ("tiny", "bf16x6"), ("mid", "bf16x6"),
                                        ("tiny", "f32"), ("mid", "f32"), ("softmax", "f32")] + [(n, "f16x3") for n in COMPOSED + STAGED] +
                         [(n, "f32") for n in STAGED])
def test_golden_forward_loss_grads(golden_dir, name, arith):
    """fused configurations in every arithmetic of the contraction; the configurations outside the fused family (causal / cLN,
    non-separable P = 5, non-dilated without norm) through the module-by-module composition on the GPU"""
    prev = sepkernels.set_gemm_arith(arith)
    try:
        _golden_case(golden_dir, name)
    finally:
        sepkernels.set_gemm_arith(prev)


def _golden_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "convtasnet_{}.npz".format(name)))
    model = ConvTasNet(**CONFIGS[name])
    assert model.fused == (name not in COMPOSED + STAGED) and model.staged == (name in STAGED) and model.fused_derived == (name in DERIVED)
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")})
    model.cuda()
    mixture, sources = torch.from_numpy(g["mixture"]).cuda(), torch.from_numpy(g["sources"]).cuda()
    est, latent = model.extract_latent(mixture)
    ref = torch.from_numpy(g["output_f64"])
    assert est.shape == ref.shape
    assert _rel(est, ref) <= TOL
    lsum = latent.sum().real.item() if torch.is_complex(latent) else latent.double().sum().item()
    assert abs(lsum - float(g["latent_f64_sum"])) <= TOL * float(g["latent_f64_abs_sum"])
    loss, pattern = PIT1d(NegSISDR(), n_sources=CONFIGS[name]["n_sources"])(est, sources)
    assert abs(loss.item() - float(g["loss_f64"])) <= TOL * abs(float(g["loss_f64"]))
    assert np.array_equal(pattern.cpu().numpy(), g["pattern"])
    loss.backward()
    ref_grads = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad/")}
    flat_rel, worst = _grad_report(model, ref_grads)
    assert flat_rel <= TOL, "flat gradient rel-inf {:.3e}, worst tensor {}".format(flat_rel, worst)
    if name in STAGED:
        # the causal family layer by layer on the kernels (models/conv_tasnet.py::_run_staged): every tensor at 1e-3 of its own scale; a
        # slope is judged on the flat vector above (cancelling sums, see test_batch16_against_the_fp64_reference_fixture)
        for k, q in model.named_parameters():
            r = ref_grads[k].double()
            rel = (q.grad.double().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-30)
            assert rel <= 1e-3 or q.numel() == 1, "{}: {:.3e}".format(k, rel)
    elif name not in COMPOSED:
        # per tensor: max(1e-3, 2 x the oracle's own fp32-vs-fp64 error on that tensor)
        noise = _oracle_fp32_noise(CONFIGS[name], g)
        for k, q in model.named_parameters():
            r = ref_grads[k].double()
            rel = (q.grad.double().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-30)
            assert rel <= max(1e-3, 2 * noise[k]), "{}: {:.3e} (oracle fp32 noise {:.3e})".format(k, rel, noise[k])
    elif name in DERIVED:
        # the fused sequence on derived bases: the separator's tensors at the fused gate; the filterbank's own parameters (a window, a
        # few dozen frequencies / phases) are sums over every frame of the batch like the slopes: flat vector only
        for k, q in model.named_parameters():
            if k not in ref_grads or not k.startswith("separator.") or q.numel() == 1:
                continue
            r = ref_grads[k].double()
            rel = (q.grad.double().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-30)
            assert rel <= 5e-3, "{}: {:.3e}".format(k, rel)
    else:
        # the module-by-module composition (torch convolutions around the library's norms): every tensor at the bar of the forward; measured
        # worst 1.5e-6 ... 4.7e-6 over the five configurations (tools/print_composed_worst.py, profiles/r05zp_composed_worst.txt)
        assert worst[1] <= 1e-3, "per-tensor worst {}".format(worst)


def test_paper_best_against_oracle():
    """Paper-best (N512 L16 B128 H512 Sc128 P3 X8 R3), 1 utterance of 4 s @ 8 kHz: the fp64 CPU port is the truth."""
    torch.manual_seed(111)
    model = ConvTasNet(**PAPER)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    sources = 0.1 * torch.randn(1, 2, 32000, generator=g)
    mixture = sources.sum(1, keepdim=True)
    ref_out, ref_loss, ref_pat, ref_grads = FP.train_step(p, PAPER, mixture, sources, dtype=torch.float64)
    model.cuda()
    est = model(mixture.cuda())
    assert _rel(est, ref_out) <= TOL
    loss, pattern = PIT1d(NegSISDR(), n_sources=2)(est, sources.cuda())
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    assert torch.equal(pattern.cpu(), ref_pat)
    loss.backward()
    flat_rel, worst = _grad_report(model, ref_grads)
    assert flat_rel <= TOL, "flat gradient rel-inf {:.3e}, worst tensor {}".format(flat_rel, worst)


def test_full_size_properties():
    """BASELINE.json configs[1] size (B=16, T=32000): properties that need no oracle."""
    torch.manual_seed(111)
    model = ConvTasNet(**PAPER).cuda()
    g = torch.Generator().manual_seed(9)
    sources = (0.1 * torch.randn(16, 2, 32000, generator=g)).cuda()
    mixture = sources.sum(1, keepdim=True)
    crit = PIT1d(NegSISDR(), n_sources=2)
    with torch.no_grad():
        est = model(mixture)
        # (1) utterances are independent: a sample processed alone gives the same answer as inside the batch
        for b in (0, 15):
            alone = model(mixture[b:b + 1])
            assert _rel(alone[0], est[b].cpu()) <= 1e-4
        # (2) linear encoder + gLN => the network is scale-equivariant (eps negligible at this level)
        est2 = model(3.0 * mixture)
        assert _rel(est2, (3.0 * est).cpu()) <= 1e-3
        # (3) PIT is invariant to relabelling the sources and returns the relabelled pattern
        loss, pattern = crit(est, sources)
        loss_sw, pattern_sw = crit(est, sources[:, [1, 0]])
        assert abs(loss.item() - loss_sw.item()) <= 1e-5 * abs(loss.item())
        assert torch.equal(pattern_sw, 1 - pattern)
    # (4) directional derivative: <grad, d> matches a central finite difference of the loss along d
    est = model(mixture)
    loss, _ = crit(est, sources)
    loss.backward()
    params = [p for p in model.parameters()]
    torch.manual_seed(3)
    dirs = [torch.randn_like(p) * p.detach().abs().mean().clamp_min(1e-3) for p in params]
    analytic = sum((p.grad.double() * d.double()).sum().item() for p, d in zip(params, dirs))
    h = 2e-3
    vals = []
    with torch.no_grad():
        for sgn in (+1, -1):
            for p, d in zip(params, dirs):
                p.add_(sgn * h * d)
            vals.append(crit(model(mixture), sources)[0].double().item())
            for p, d in zip(params, dirs):
                p.sub_(sgn * h * d)
    numeric = (vals[0] - vals[1]) / (2 * h)
    assert abs(analytic - numeric) <= 3e-2 * abs(numeric) + 1e-3, (analytic, numeric)


def test_batch16_each_utterance_against_the_oracle():
    """BASELINE configs[1] at its real batch: ONE B = 16 forward on the GPU against ONE run of the oracle on the same batch (CPU port, pinned
    to the live reference at this configuration by tests/test_oracle_vs_reference_cpu.py; fp32: its forward is good to 1e-6): every
    utterance's outputs to 1e-3, its PIT loss and permutation, and the batch loss, on utterances and speakers of different level.  The
    GRADIENTS of this configuration and batch are judged against the reference itself in fp64 by the fixture test below (round-4 verdict:
    the fp32-vs-fp32 gradient gates this test used to carry -- 2e-3 flat, 1e-2 per tensor -- added nothing beside it and are gone)."""
    B, T = 16, 32000
    torch.manual_seed(111)
    model = ConvTasNet(**PAPER)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, q in model.named_parameters():
            if n.endswith("norm.weight") or n.endswith("norm.bias"):
                q.add_(0.1 * torch.randn(q.shape, generator=g))
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sources = 0.1 * torch.randn(B, 2, T, generator=g) * torch.exp(0.7 * torch.randn(B, 2, 1, generator=g))     # utterances / speakers of different level
    mixture = sources.sum(1, keepdim=True)
    model.cuda()
    crit = PIT1d(NegSISDR(), n_sources=2)
    with torch.no_grad():
        est = model(mixture.cuda())
        loss, pattern = crit(est, sources.cuda())
        per_utt, _ = crit(est, sources.cuda(), batch_mean=False)
        o, l, pat = FP.forward_loss(p, PAPER, mixture, sources, dtype=torch.float32)
    assert abs(loss.item() - l.item()) <= TOL * max(abs(l.item()), 1.0)
    assert torch.equal(pattern.cpu(), torch.as_tensor(pat))
    for b in range(B):
        assert _rel(est[b], o[b]) <= TOL, b
        lb, _ = FP.neg_sisdr_pit(o[b:b + 1].double(), sources[b:b + 1].double())
        assert abs(per_utt[b].item() - lb.item()) <= TOL * max(abs(lb.item()), 1.0), b


def test_batch16_against_the_fp64_reference_fixture(golden_dir):
    """BASELINE configs[1] at its real batch against the REFERENCE ITSELF in fp64: tests/golden/convtasnet_paper_b16.npz was written in the
    build container by oracle/make_golden.py::paper_b16_golden from the unmodified reference classes (172 s of host time) -- outputs, the
    sixteen PIT losses and permutations, and a fingerprint (norm, largest magnitude, 64 sampled elements) of every parameter gradient.
    Gates: outputs 1e-3 per utterance, losses 1e-3, and EVERY gradient tensor -- the 49 scalar PReLU slopes included -- flat 1e-3 of the
    tensor's own largest magnitude on the sampled elements and 1e-3 on its norm."""
    from oracle.make_golden import PAPER_CFG, PAPER_B16_SEEDS, paper_b16_inputs, sample_index
    fx = np.load(os.path.join(golden_dir, "convtasnet_paper_b16.npz"))
    torch.manual_seed(PAPER_B16_SEEDS["model"])
    model = ConvTasNet(**PAPER_CFG)
    mixture, sources = paper_b16_inputs(model)
    assert np.array_equal(mixture.numpy()[:, 0, :8], fx["mixture_head"])
    for k, v in model.state_dict().items():      # same parameters as the reference drew (default init under the seed + the perturbation)
        want = fx["pfp/" + k]
        got = np.array([v.double().sum().item(), v.double().abs().sum().item()])
        assert np.allclose(got, want, rtol=1e-9, atol=1e-12), k
    model.cuda()
    est = model(mixture.cuda())
    crit = PIT1d(NegSISDR(), n_sources=2)
    loss, pattern = crit(est, sources.cuda())
    per_utt, _ = crit(est.detach(), sources.cuda(), batch_mean=False)
    loss.backward()
    o = est.detach().double().cpu()
    idx = fx["out_idx"].tolist()
    amax = torch.as_tensor(fx["out_amax"])
    assert ((o[:, :, idx] - torch.as_tensor(fx["out_samples"])).abs().amax(2) <= TOL * amax).all()
    assert ((o.norm(dim=2) - torch.as_tensor(fx["out_norm"])).abs() <= TOL * torch.as_tensor(fx["out_norm"])).all()
    assert np.array_equal(pattern.cpu().numpy(), fx["pattern"])
    assert abs(loss.item() - float(fx["loss_f64"])) <= TOL * abs(float(fx["loss_f64"]))
    assert (np.abs(per_utt.double().cpu().numpy() - fx["per_utt_f64"]) <= TOL * np.abs(fx["per_utt_f64"])).all()
    # gradients.  Three gates:
    #   flat      every sampled element of every tensor (slopes included) within 1e-3 of the LARGEST gradient entry of the model;
    #   tensor    every non-scalar tensor within 1e-3 of its own largest entry (sampled elements) and 1e-3 on its norm;
    #   slopes    the 49 scalar PReLU slopes as one vector: within 1e-3 of the largest slope gradient.  A single slope is a signed sum over
    #             33 M terms that can cancel to 1e-6 of their magnitude: against its OWN value no fp32 evaluation is good to 1e-3 (the
    #             reference's fp32 run misses one of them by 8 % on this batch), so the slopes are judged on their common scale.
    gmax = max(float(fx["gfp/" + k][1]) for k, _ in model.named_parameters())
    slopes = [k for k, q in model.named_parameters() if q.numel() == 1]
    smax = max(float(fx["gfp/" + k][1]) for k in slopes)
    worst_flat, worst_tensor, worst_slope, worst_slope_own = (0.0, None), (0.0, None), (0.0, None), (0.0, None)
    for k, q in model.named_parameters():
        fp = fx["gfp/" + k]
        g = q.grad.double().cpu().reshape(-1)
        n64, a64, s64 = fp[0], fp[1], torch.as_tensor(fp[2:])
        diff = (g[sample_index(g.numel())] - s64).abs().max().item()
        worst_flat = max(worst_flat, (diff / gmax, k))
        if q.numel() == 1:
            worst_slope = max(worst_slope, (diff / smax, k))
            worst_slope_own = max(worst_slope_own, (diff / (a64 + 1e-300), k))
        else:
            worst_tensor = max(worst_tensor, (max(diff / (a64 + 1e-300), abs(g.norm().item() - n64) / (n64 + 1e-300)), k))
    print("batch-16 gradients vs the fp64 reference: flat {:.2e} ({}), worst tensor {:.2e} ({}), slopes as a vector {:.2e} ({}), worst slope against "
          "its own value {:.2e} ({})".format(worst_flat[0], worst_flat[1], worst_tensor[0], worst_tensor[1], worst_slope[0], worst_slope[1],
                                            worst_slope_own[0], worst_slope_own[1]))
    assert worst_flat[0] <= TOL, worst_flat
    assert worst_tensor[0] <= TOL, worst_tensor
    assert worst_slope[0] <= TOL, worst_slope


def test_f16x3_model_with_adversarial_weight_scales():
    """SEP_ARITH_F16X3 with the weights as hostile to a shared scale as they get: one layer's 1x1 weights scaled up by 2^20 (its
    bias and the following norm absorb it: gLN is scale invariant), another layer's scaled down by 2^-20, one single huge entry in a
    third.  The packer scales every ROW of every matrix on its own, so the output must still match the fp64 oracle to 1e-3 -- with one
    bound for all weights (round 1) the small layers lose all their bits."""
    cfg = CONFIGS["mid"]
    torch.manual_seed(3)
    model = ConvTasNet(**cfg)
    with torch.no_grad():
        sd = model.state_dict()
        sd["separator.tdcn.net.0.net.0.bottleneck_conv1d.weight"].mul_(2.0 ** 20)
        sd["separator.tdcn.net.0.net.0.bottleneck_conv1d.bias"].mul_(2.0 ** 20)
        sd["separator.tdcn.net.1.net.1.bottleneck_conv1d.weight"].mul_(2.0 ** -20)
        sd["separator.tdcn.net.1.net.1.bottleneck_conv1d.bias"].mul_(2.0 ** -20)
        sd["separator.tdcn.net.0.net.2.separable_conv1d.skip_pointwise_conv1d.weight"][3, 5, 0] = 3.0e4
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sources = 0.1 * torch.randn(2, cfg["n_sources"], 3000, generator=torch.Generator().manual_seed(4))
    mixture = sources.sum(1, keepdim=True)
    ref_out, ref_loss, ref_pat, ref_grads = FP.train_step(p, cfg, mixture, sources, dtype=torch.float64)
    model.cuda()
    est = model(mixture.cuda())
    assert _rel(est.detach(), ref_out) <= TOL
    loss, pattern = PIT1d(NegSISDR(), n_sources=cfg["n_sources"])(est, sources.cuda())
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    loss.backward()
    # gradients of the rescaled layers live at 2^-20 / 2^+20 of the others: judge every tensor against its own scale
    for k, q in model.named_parameters():
        r = ref_grads[k].double()
        rel = (q.grad.double().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-30)
        assert rel <= 5e-3, "{}: {:.3e}".format(k, rel)


def _run_steps(cfg, batches, recorded, lr_change_at=None):
    from sepkernels.train import FusedTrainStep
    torch.manual_seed(1)
    model = ConvTasNet(**cfg).cuda()
    step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=cfg["n_sources"]), lr=1e-3, max_norm=5.0, auto_record=recorded)
    losses = []
    for i, src in enumerate(batches):
        mix = src.sum(1, keepdim=True).contiguous()
        if i == lr_change_at:
            step.lr = 5e-4
        losses.append(step(mix, src).item())
    torch.cuda.synchronize()
    assert step.step_count == len(batches)
    assert (step._seq is not None) == recorded
    return losses, model.flat_parameters().detach().clone(), step


def test_recorded_step_equals_eager_steps():
    """FusedTrainStep with auto_record: the first step is recorded (sepkernels.Sequence), the other five are ONE sep_run_sequence call each
    (with a learning-rate change before the last one: the rate lives in device memory) -- against six plain eager steps from the same
    initial state.  Same entry points, same arguments, same order: the losses agree to rounding of the loss's own batch mean."""
    cfg = CONFIGS["mid"]
    g = torch.Generator().manual_seed(9)
    batches = [0.1 * torch.randn(2, cfg["n_sources"], 3203, generator=g).cuda() for _ in range(6)]
    (l0, p0, _), (l1, p1, step) = _run_steps(cfg, batches, False, 5), _run_steps(cfg, batches, True, 5)
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-6 * abs(a), (l0, l1)
    assert (p0 - p1).abs().max().item() <= 1e-6 * p0.abs().max().item()
    names = step._seq.names()
    assert names[0] in ("sep_absmax", "sep_memset") and names[-1] == "sep_adam_step_dev" and "sep_pit_finish" in names
    # a batch of another shape steps eagerly and keeps the device-side step count in line; the recorded shape replays again afterwards
    other = 0.1 * torch.randn(1, cfg["n_sources"], 2000, generator=g).cuda()
    step(other.sum(1, keepdim=True).contiguous(), other)
    step(batches[0].sum(1, keepdim=True).contiguous(), batches[0])
    torch.cuda.synchronize()
    assert step.step_count == 8 and int(step._step_dev.item()) == 8


def test_recorded_step_at_paper_best_sixteen_utterances_trains_like_the_eager_step():
    """BASELINE configs[1] at the bench's own size: ten steps as recording + nine replays against ten eager steps, every loss compared.
    (hipGraph replay of this step ended with inf / 49.98 losses in half the runs on this stack, profiles/r07_round5_experiments.md r07m:
    that path is gone; this is the replacement's proof at product size.)"""
    g = torch.Generator().manual_seed(111)
    src = (0.1 * torch.randn(16, 2, 32000, generator=g)).cuda()
    batches = [src] * 10
    l0, p0, _ = _run_steps(PAPER, batches, False)
    torch.cuda.empty_cache()
    l1, p1, _ = _run_steps(PAPER, batches, True)
    assert l0[-1] < l0[0] - 1.0                                      # (the steps did train)
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (l0, l1)
    # the two runs differ by the order of the statistics' fp64 atomics (rounding of a few gradients); Adam's normalised update turns a
    # gradient that IS rounding noise into a step of either sign, so a handful of parameters may sit up to 2 lr per step apart
    diff = (p0 - p1).abs()
    assert (diff > 1e-5).float().mean().item() <= 1e-3 and diff.max().item() <= 2.1e-2, ((diff > 1e-5).float().mean().item(), diff.max().item())


def test_recorded_sinkpit_step_equals_eager_steps():
    """BASELINE configs[4]'s criterion in the recorded step: SinkPIT(NegSI-SDR, coldness 1, 20 iterations) on a 4-speaker softmax-mask model,
    recording + three replays against four eager steps (sep_axpby for C = -SI-SDR and dL/dSI-SDR = -dL/dC, sep_pit_finish for the batch mean)."""
    from sepkernels.train import FusedTrainStep
    cfg = dict(CONFIGS["mid"], n_sources=4, mask_nonlinear="softmax")
    g = torch.Generator().manual_seed(4)
    batches = [0.1 * torch.randn(3, 4, 3203, generator=g).cuda() for _ in range(4)]
    runs = []
    for recorded in (False, True):
        torch.manual_seed(1)
        model = ConvTasNet(**cfg).cuda()
        step = FusedTrainStep(model, SinkPIT(NegSISDR(), n_sources=4, coldness=1.0, iteration=20), lr=1e-3, max_norm=5.0, auto_record=recorded)
        losses = [step(src.sum(1, keepdim=True).contiguous(), src).item() for src in batches]
        torch.cuda.synchronize()
        assert (step._seq is not None) == recorded and step.step_count == 4
        if recorded:
            assert "sep_sinkhorn_fwd" in step._seq.names() and step.last_pattern.shape == (3, 4)
        runs.append((losses, model.flat_parameters().detach().clone()))
    (l0, p0), (l1, p1) = runs
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-5 * abs(a), (l0, l1)
    assert (p0 - p1).abs().max().item() <= 1e-5 * p0.abs().max().item()


def test_gated_encoder_on_the_device():
    """models.filterbank.GatedEncoder on sep_encoder_fwd / sep_unfold + sep_pw_wgrad against nn.Conv1d in fp64 (reference filterbank.py:325-346)"""
    from models.filterbank import GatedEncoder
    torch.manual_seed(4)
    enc = GatedEncoder(1, 512, kernel_size=16, stride=8)
    ref = GatedEncoder(1, 512, kernel_size=16, stride=8).double()
    ref.load_state_dict({k: v.double() for k, v in enc.state_dict().items()})
    x = torch.randn(4, 1, 16 + 8 * 3998)
    w = torch.randn(4, 512, 3999)
    yr = ref(x.double())                                            # CPU tensors: the convolutions
    rU, rV = torch.autograd.grad((yr * w.double()).sum(), [ref.conv1d_U.weight, ref.conv1d_V.weight])
    enc.cuda()
    assert enc._on_kernels(x.cuda())
    y = enc(x.cuda())
    gU, gV = torch.autograd.grad((y * w.cuda()).sum(), [enc.conv1d_U.weight, enc.conv1d_V.weight])
    assert y.shape == (4, 512, 3999) and _rel(y, yr.detach()) <= 1e-5
    assert _rel(gU, rU) <= 1e-4 and _rel(gV, rV) <= 1e-4


def test_record_refuses_what_it_does_not_implement():
    from sepkernels.train import FusedTrainStep
    from criterion.sdr import ClippedNegSISDR
    model = ConvTasNet(**CONFIGS["tiny"]).cuda()
    src = 0.1 * torch.randn(2, 2, 2000).cuda()
    step = FusedTrainStep(model, PIT1d(ClippedNegSISDR(min=-30), n_sources=2), auto_record=True)
    assert "over SI-SDR" in step.recordable()
    with pytest.raises(RuntimeError, match="over SI-SDR"):
        step.record(src.sum(1, keepdim=True), src)
    step(src.sum(1, keepdim=True).contiguous(), src)                  # auto_record falls back to the eager step
    assert step._seq is None and step.step_count == 1


def test_multichannel_relu_encoder_and_validation_length():
    """in_channels=2 (4-D input), enc ReLU, a length that needs input padding, B=1 (the validation/test regime)."""
    cfg = dict(CONFIGS["tiny"], in_channels=2, n_sources=3)
    torch.manual_seed(1)
    model = ConvTasNet(**cfg)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = 0.1 * torch.randn(1, 1, 2, 5003)
    ref_out, _ = FP.conv_tasnet(x.view(1, 2, 5003).double(), {k: v.double() for k, v in p.items()}, cfg)
    model.cuda()
    with torch.no_grad():
        out = model(x.cuda())
    assert out.shape == (1, 3, 2, 5003)
    # reference view semantics: decoder output (B*n_src, in_channels, T) -> (B, n_src, n_mics, T)
    F_ref = torch.nn.functional
    assert _rel(out.view(1, 3, 2, 5003), ref_out.view(1, 3, 2, 5003)) <= TOL


def test_criteria_on_gpu(golden_dir):
    g = np.load(os.path.join(golden_dir, "pit_kat.npz"))
    x, t = torch.from_numpy(g["pit_x"]).cuda(), torch.from_numpy(g["pit_t"]).cuda()
    loss, pattern = PIT1d(SISDR(), n_sources=2)(x, t)
    assert abs(loss.item() - (-4.6058)) < 1e-3 and pattern.tolist() == [[1, 0], [1, 0], [0, 1], [0, 1]]
    x, t = torch.from_numpy(g["sink_x"]).cuda(), torch.from_numpy(g["sink_t"]).cuda()
    loss, pattern = PIT1d(NegSISDR(), n_sources=3)(x, t)
    assert abs(loss.item() - 4.4252) < 1e-3 and pattern.tolist() == [[1, 0, 2], [2, 1, 0], [0, 1, 2], [2, 1, 0]]
    loss, pattern = SinkPIT(NegSISDR(), n_sources=3, coldness=1)(x, t, batch_mean=False)
    assert np.allclose(loss.cpu().numpy(), [11.1611, 10.4200, 10.5582, 9.9728], atol=1e-3)
    assert pattern.tolist() == [[2, 0, 2], [0, 1, 0], [0, 1, 2], [2, 1, 0]]
    x = torch.from_numpy(g["sg_x"]).float().cuda().requires_grad_(True)
    t = torch.from_numpy(g["sg_t"]).float().cuda()
    loss, pattern = SinkPIT(NegSISDR(), n_sources=4, coldness=2.0, iteration=20)(x, t)
    loss.backward()
    assert abs(loss.item() - float(g["sg_loss"])) < 1e-4 * abs(float(g["sg_loss"])) + 1e-4
    assert _rel(x.grad, torch.from_numpy(g["sg_grad"])) <= 1e-3
    assert np.array_equal(pattern.cpu().numpy(), g["sg_pattern"])


def test_global_layer_norm_module_on_gpu():
    torch.manual_seed(0)
    x = (torch.randn(3, 24, 777) * 2 + 0.3)
    norm = GlobalLayerNorm(24)
    with torch.no_grad():
        norm.norm.weight.add_(0.3 * torch.randn(24)); norm.norm.bias.add_(0.2 * torch.randn(24))
    ref = torch.nn.GroupNorm(1, 24, eps=1e-12).double()
    ref.load_state_dict({k: v.double() for k, v in norm.norm.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (yr ** 2).sum().backward()
    norm.cuda()
    xg = x.cuda().requires_grad_(True)
    y = norm(xg)
    (y ** 2).sum().backward()
    assert _rel(y, yr.detach()) <= 1e-4
    assert _rel(xg.grad, xr.grad) <= 1e-3
    assert _rel(norm.norm.weight.grad, ref.weight.grad) <= 1e-3
    assert _rel(norm.norm.bias.grad, ref.bias.grad) <= 1e-3


def test_dprnn_tasnet_golden(golden_dir):
    """BASELINE.json configs[3] family (small instance): head/tail kernels + segment/overlap-add + the LSTM sweep kernels."""
    from oracle.make_golden import DPRNN_CFG
    from models.dprnn_tasnet import DPRNNTasNet
    g = np.load(os.path.join(golden_dir, "dprnn_tasnet_small.npz"))
    model = DPRNNTasNet(**DPRNN_CFG)
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")})
    model.cuda()
    mixture, sources = torch.from_numpy(g["mixture"]).cuda(), torch.from_numpy(g["sources"]).cuda()
    est, latent = model.extract_latent(mixture)
    ref = torch.from_numpy(g["output_f64"])
    assert _rel(est, ref) <= TOL
    loss, pattern = PIT1d(NegSISDR(), n_sources=2)(est, sources)
    assert abs(loss.item() - float(g["loss_f64"])) <= TOL * abs(float(g["loss_f64"]))
    assert np.array_equal(pattern.cpu().numpy(), g["pattern"])
    loss.backward()
    flat_rel, worst = _grad_report(model, {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad/")})
    assert flat_rel <= 2e-3, "flat gradient rel-inf {:.3e}, worst tensor {}".format(flat_rel, worst)


def _dprnn_full_model_and_data(g):
    """the product's DPRNN-TasNet at BASELINE configs[3]'s real size under the seeds of oracle/make_golden.py::dprnn_full_golden; the 2.6 M
    parameters are not in the fixture, so their fingerprints (sum, abs-sum per tensor) are checked against the reference's first"""
    from oracle.make_golden import DPRNN_FULL_CFG, DPRNN_FULL_SEEDS, perturb
    from models.dprnn_tasnet import DPRNNTasNet
    torch.manual_seed(DPRNN_FULL_SEEDS["model"])
    model = DPRNNTasNet(**DPRNN_FULL_CFG)
    perturb(model, DPRNN_FULL_SEEDS["perturb"])
    for k, v in model.state_dict().items():
        fp = g["pfp/" + k]
        assert abs(v.double().sum().item() - fp[0]) <= 1e-6 * (fp[1] + 1e-12) and abs(v.double().abs().sum().item() - fp[1]) <= 1e-6 * (fp[1] + 1e-12), k
    gen = torch.Generator().manual_seed(DPRNN_FULL_SEEDS["data"])
    sources = 0.1 * torch.randn(1, 2, 32000, generator=gen)
    mixture = sources.sum(dim=1, keepdim=True)
    assert np.array_equal(mixture.numpy()[0, 0, :16], g["mixture_head"])
    return model, mixture, sources


def test_dprnn_tasnet_full_size_golden(golden_dir):
    """BASELINE.json configs[3] at its REAL size (N=64, L=2, F=64, H=128, K=250, P=125, 6 blocks; one utterance of 4 s @ 8 kHz = 31 999
    frames, 255 chunks) against the unmodified reference's fp64 run (tests/golden/dprnn_tasnet_full.npz, written by
    oracle/make_golden.py::dprnn_full_golden): output, PIT loss, permutation and every parameter gradient through its fingerprint (L2 norm
    and 64 sampled elements per tensor)."""
    from oracle.make_golden import sample_index
    g = np.load(os.path.join(golden_dir, "dprnn_tasnet_full.npz"))
    model, mixture, sources = _dprnn_full_model_and_data(g)
    model.cuda()
    est = model(mixture.cuda())
    ref = torch.from_numpy(g["output_f64"])
    assert _rel(est, ref) <= TOL
    loss, pattern = PIT1d(NegSISDR(), n_sources=2)(est, sources.cuda())
    assert abs(loss.item() - float(g["loss_f64"])) <= TOL * abs(float(g["loss_f64"]))
    assert np.array_equal(pattern.cpu().numpy(), g["pattern"])
    loss.backward()
    gmax = max(float(g["gfp/" + k][1]) for k, _ in model.named_parameters())
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        fp = g["gfp/" + k]
        gr = p.grad.detach().double().cpu().reshape(-1)
        assert abs(gr.norm().item() - fp[0]) <= 2e-3 * fp[0] + 1e-6 * gmax, (k, gr.norm().item(), fp[0])
        e = (gr[sample_index(gr.numel())] - torch.from_numpy(fp[2:]).double()).abs().max().item() / gmax
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] <= 2e-3, "sampled gradient elements: rel-inf {:.3e} (of the largest gradient magnitude) in {}".format(worst[1], worst[0])


@pytest.mark.parametrize("name", ["dptnet", "dptnet_causal", "dptnet_odd", "galrnet", "galrnet_causal", "sepformer", "sepformer_causal",
                                  "dprnn_tasnet_causal", "dprnn_tasnet_odd", "dprnn_tasnet_softmax"])
def test_sibling_separators_golden(golden_dir, name):
    """DPTNet / GALRNet / SepFormer on the device against the reference's fp64 run -- encoder, 1x1
    convolutions (bottleneck, PReLU + map, the stacked GTU pair, SepFormer's fused head and output convolution), chunking,
    gLN / cLN, LSTM sweeps and mask * w + decoder on libsepkernels; `_odd` (widths off the multiples of 16, softmax mask) is the
    composition path on the device, and so are DPRNN-TasNet's causal / odd-width instances (`dprnn_tasnet_softmax`: the head / tail
    kernels with the channel softmax)."""
    from oracle.make_golden import SIBLINGS
    from models.dptnet import DPTNet
    from models.galrnet import GALRNet
    from models.sepformer import SepFormer
    from models.dprnn_tasnet import DPRNNTasNet
    kind, cfg = SIBLINGS[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    model = {"DPTNet": DPTNet, "GALRNet": GALRNet, "SepFormer": SepFormer, "DPRNNTasNet": DPRNNTasNet}[kind](**cfg)
    assert list(model.state_dict().keys()) == list(g["state_keys"])
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}, strict=False)
    model.cuda()
    assert (not model.kernel_path_problems()) == (name not in ("dptnet_odd", "dprnn_tasnet_odd", "dprnn_tasnet_causal"))
    mixture, sources = torch.from_numpy(g["mixture"]).cuda(), torch.from_numpy(g["sources"]).cuda()
    est, latent = model.extract_latent(mixture)
    assert _rel(est, torch.from_numpy(g["output_f64"])) <= TOL
    assert abs(latent.sum().item() - float(g["latent_f64_sum"])) <= TOL * float(g["latent_f64_abs_sum"])
    loss, pattern = PIT1d(NegSISDR(), n_sources=cfg["n_sources"])(est, sources)
    assert abs(loss.item() - float(g["loss_f64"])) <= TOL * abs(float(g["loss_f64"]))
    assert np.array_equal(pattern.cpu().numpy(), g["pattern"])
    loss.backward()
    flat_rel, worst = _grad_report(model, {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad/")})
    assert flat_rel <= 2e-3, "flat gradient rel-inf {:.3e}, worst tensor {}".format(flat_rel, worst)


@pytest.mark.parametrize("widths", [(48, 48, 80, 16), (80, 16, 48, 48)])
def test_widths_in_odd_multiples_of_16_against_oracle(widths):
    """N / B / H / Sc in multiples of 16 that are not multiples of 32: the weight packer works in 32-row blocks, those products take the
    fp32 weights (the step used to die in sep_pack_weights; tests/test_modules_cpu.py::test_widths_in_odd_multiples_of_16)."""
    N, Bn, H, Sc = widths
    cfg = dict(n_basis=N, kernel_size=8, stride=4, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu", sep_hidden_channels=H,
               sep_bottleneck_channels=Bn, sep_skip_channels=Sc, sep_kernel_size=3, sep_num_blocks=2, sep_num_layers=2, dilated=True, separable=True,
               causal=False, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear="sigmoid", n_sources=2)
    torch.manual_seed(3)
    model = ConvTasNet(**cfg)
    assert model.fused
    sources = 0.1 * torch.randn(2, 2, 700)
    mixture = sources.sum(1, keepdim=True)
    p64 = {k: v.detach().double() for k, v in model.state_dict().items()}
    ref_out, ref_loss, ref_pat, ref_grads = FP.train_step(p64, cfg, mixture, sources, dtype=torch.float64)
    model.cuda()
    est = model(mixture.cuda())
    assert _rel(est, ref_out) <= TOL
    loss, pattern = PIT1d(NegSISDR(), n_sources=2)(est, sources.cuda())
    assert torch.equal(pattern.cpu(), ref_pat) and abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    loss.backward()
    flat_rel, worst = _grad_report(model, ref_grads)
    assert flat_rel <= 2e-3, "flat gradient rel-inf {:.3e}, worst tensor {}".format(flat_rel, worst)


def test_sepformer_encoder_stack_token_route_against_torch_layers_and_with_dropout():
    """SepFormer's intra-chunk stack at its recipe widths (256 features, 8 heads, 1024 hidden: reference sepformer.py:395-470) on the device:
    the token-major route (csrc/linear.hip, csrc/attn.hip, the feed-forward pair on the convolution kernels, residual sum + LayerNorm on
    sep_rownorm_*) against torch's own nn.TransformerEncoder on the strided route in float64 (evaluation mode); then training mode with the
    recipe's dropout 0.1: finite, different from the evaluation output by about what 10 % dropout does, and every parameter gets a gradient."""
    from models.sepformer import IntraTransformer
    torch.manual_seed(21)
    net = IntraTransformer(256, num_layers=2, num_heads=8, d_ff=1024, norm=True, dropout=0.1)
    x = 0.5 * torch.randn(2, 256, 5, 50)
    ref_net = IntraTransformer(256, num_layers=2, num_heads=8, d_ff=1024, norm=True, dropout=0.1).double()
    ref_net.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    ref_net.eval()
    ref_net._tokens_ok = lambda t: False                                      # torch's layers, module by module
    x64 = x.double().requires_grad_(True)
    ref = ref_net(x64)
    w = torch.randn(x.shape)
    (ref * w.double()).sum().backward()
    net.cuda().eval()
    xd = x.cuda().requires_grad_(True)
    assert net._tokens_ok(xd)
    y = net(xd)
    (y * w.cuda()).sum().backward()
    assert _rel(y, ref.detach()) <= TOL and _rel(xd.grad, x64.grad) <= 2e-3
    for (k, p), q in zip(net.named_parameters(), ref_net.parameters()):
        assert _rel(p.grad, q.grad) <= 2e-3, k
    net.train()
    net.zero_grad()
    torch.manual_seed(5)
    yt = net(xd)
    assert torch.isfinite(yt).all()
    d = (yt - y).abs().mean().item() / y.abs().mean().item()
    assert 0.01 < d < 1.0, d
    (yt * w.cuda()).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0 for p in net.parameters())
    torch.manual_seed(5)
    assert torch.equal(net(xd), yt)                                           # the masks are functions of the host generator's state


def test_dptnet_full_width_step():
    """DPTNet at the widths of its paper (N = 64, L = 2, 64 bottleneck channels, chunks of 100 frames) on 4 x 2 s @ 8 kHz:
    320 chunks per utterance, i.e. 1280 x 64 rows in every intra-chunk gLN -- more than one launch takes (the module splits the
    batch); LSTM sweeps at H = 128.  Size-independent properties: finite loss / gradients, utterances independent of each other."""
    from models.dptnet import DPTNet
    torch.manual_seed(3)
    model = DPTNet(64, 2, stride=1, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu", sep_bottleneck_channels=64,
                   sep_hidden_channels=128, sep_chunk_size=100, sep_num_blocks=2, sep_num_heads=4, sep_dropout=0, mask_nonlinear="relu",
                   causal=False, n_sources=2).cuda()
    assert not model.kernel_path_problems()
    g = torch.Generator().manual_seed(12)
    sources = (0.1 * torch.randn(4, 2, 16000, generator=g)).cuda()
    mixture = sources.sum(1, keepdim=True)
    est = model(mixture)
    assert est.shape == (4, 2, 16000) and torch.isfinite(est).all()
    loss, _ = PIT1d(NegSISDR(), n_sources=2)(est, sources)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    with torch.no_grad():
        alone = model(mixture[2:3])
    assert _rel(alone, est[2:3].detach().cpu()) <= 1e-4


def test_orpit_and_dsconv_on_gpu(golden_dir):
    from test_modules_cpu import _orpit_case, _dsconv_case
    g, x, loss, idx = _orpit_case(golden_dir, "cuda")
    assert np.allclose(loss.detach().cpu().numpy(), g["orpit_loss"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(idx.cpu().numpy(), g["orpit_idx"])
    assert _rel(x.grad, torch.from_numpy(g["orpit_grad"])) <= 1e-3
    mod, xg, y, ref_dw, ref_pw, xr, yr = _dsconv_case("cuda", torch.float32)
    assert _rel(y, yr.detach()) <= 1e-4
    assert _rel(xg.grad, xr.grad) <= 1e-3
    assert _rel(mod.depthwise_conv1d.weight.grad, ref_dw.weight.grad) <= 1e-3
    assert _rel(mod.pointwise_conv1d.weight.grad, ref_pw.weight.grad) <= 1e-3
    assert _rel(mod.pointwise_conv1d.bias.grad, ref_pw.bias.grad) <= 1e-3


def test_dprnn_tasnet_config4_full_size():
    """BASELINE.json configs[3] at its real size: N=64, L=2, F=64, H=128, K=250, P=125, 6 blocks, 2 utterances of 4 s @ 8 kHz
    (255 chunks of 250 frames).  Size-independent properties: the LSTM sweeps against nn.LSTM on the same parameters
    (MIOpen here is the independent implementation), scale handling of SI-SDR/PIT, per-utterance independence."""
    from models.dprnn_tasnet import DPRNNTasNet
    torch.manual_seed(5)
    model = DPRNNTasNet(n_basis=64, kernel_size=2, stride=1, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                        sep_hidden_channels=128, sep_bottleneck_channels=64, sep_chunk_size=250, sep_hop_size=125,
                        sep_num_blocks=6, sep_norm=True, mask_nonlinear="sigmoid", causal=False, rnn_type="lstm", n_sources=2).cuda()
    g = torch.Generator().manual_seed(9)
    sources = (0.1 * torch.randn(2, 2, 32000, generator=g)).cuda()
    mixture = sources.sum(1, keepdim=True)
    est = model(mixture)
    assert est.shape == (2, 2, 32000) and torch.isfinite(est).all()
    crit = PIT1d(NegSISDR(), n_sources=2)
    loss, pattern = crit(est, sources)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    # utterances are independent: running the second one alone gives the same estimate
    est1 = model(mixture[1:2])
    assert _rel(est1.detach(), est[1:2].detach().cpu()) <= 1e-4
    # the first intra-chunk bi-LSTM against MIOpen on the same parameters and input
    blk = model.separator.dprnn.net[0].intra_chunk_block
    x = torch.randn(510, 250, 64, device="cuda")
    from sepkernels.functional import lstm_bidirectional
    ours = lstm_bidirectional(x, blk.rnn)
    ref, _ = blk.rnn(x)
    assert _rel(ours.detach(), ref.detach().cpu()) <= 1e-5
    # ... and its gradients (input, the four weight matrices, the biases: sep_lstm_bwd + csrc/linear.hip, the recurrent weights' gradient
    # reading h one step off) against autograd through MIOpen; then the Linear behind it against F.linear
    go = torch.randn(510, 250, 256, device="cuda")
    grads = []
    for fn in (lambda t: lstm_bidirectional(t, blk.rnn), lambda t: blk.rnn(t)[0]):
        xr = x.clone().requires_grad_(True)
        blk.rnn.zero_grad()
        fn(xr).backward(go)
        grads.append([xr.grad.clone()] + [p.grad.clone() for p in blk.rnn.parameters()])
    for a, b in zip(*grads):
        assert _rel(a, b.cpu()) <= 2e-4
    from sepkernels.functional import linear_apply
    hcat = torch.randn(510 * 250, 256, device="cuda")
    grads = []
    for fn in (lambda t: linear_apply(t, blk.fc), lambda t: blk.fc(t)):
        hr = hcat.clone().requires_grad_(True)
        blk.fc.zero_grad()
        y = fn(hr)
        y.backward(torch.ones_like(y) * 0.5 + y.detach())
        grads.append([y.detach().clone(), hr.grad.clone(), blk.fc.weight.grad.clone(), blk.fc.bias.grad.clone()])
    for a, b in zip(*grads):
        assert _rel(a, b.cpu()) <= 2e-4


def test_paper_best_four_speakers_sinkpit_full_size():
    """BASELINE.json configs[4]: paper-best Conv-TasNet, 4 speakers, Sinkhorn-PIT (k = 200 iterations), 4 s @ 8 kHz, per-GPU
    batch; finite loss / gradients, permutation found = the one planted in the data, hard-PIT agreement."""
    from criterion.pit import SinkPIT
    from models.conv_tasnet import ConvTasNet
    torch.manual_seed(7)
    model = ConvTasNet(512, 16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None, sep_hidden_channels=512,
                       sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3, sep_num_blocks=3, sep_num_layers=8,
                       dilated=True, separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear="sigmoid",
                       n_sources=4).cuda()
    g = torch.Generator().manual_seed(11)
    B = 4
    sources = (0.1 * torch.randn(B, 4, 32000, generator=g)).cuda()
    mixture = sources.sum(1, keepdim=True)
    est = model(mixture)
    assert est.shape == (B, 4, 32000)
    crit = SinkPIT(NegSISDR(), n_sources=4, coldness=1.0, iteration=200)
    loss, pattern = crit(est, sources)
    assert torch.isfinite(loss) and pattern.shape == (B, 4)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    # on estimates that ARE permuted targets the Sinkhorn assignment must recover the permutation, like hard PIT does
    perm = torch.tensor([2, 0, 3, 1], device="cuda")
    fake = sources[:, perm] + 0.01 * torch.randn(B, 4, 32000, generator=g).cuda()
    _, pat_s = SinkPIT(NegSISDR(), n_sources=4, coldness=10.0, iteration=200)(fake, sources)
    _, pat_h = PIT1d(NegSISDR(), n_sources=4)(fake, sources)
    assert torch.equal(pat_s.cpu(), pat_h.cpu())


@pytest.mark.parametrize("mask", ["softmax", "sigmoid"])
def test_paper_best_four_speakers_sinkpit_against_oracle(mask):
    """BASELINE.json configs[4] at one utterance: forward, Sinkhorn-PIT (k = 200) loss and every parameter gradient against the
    fp64 CPU port (oracle/fast_port.py forward + oracle/convtasnet_oracle.py sinkpit).  "softmax" is the configuration exactly as
    bench.py --config sinkpit4 runs it (egs/tutorials/sinkpit_conv-tasnet/train.sh: mask_nonlinear softmax over all 4 x 512 channels of
    a frame: sep_softmax_ch_fwd / bwd at C = 2048)."""
    from criterion.pit import SinkPIT
    from oracle import convtasnet_oracle as O
    cfg = dict(PAPER, n_sources=4, mask_nonlinear=mask)
    torch.manual_seed(23)
    model = ConvTasNet(**cfg)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(19)
    sources = 0.1 * torch.randn(1, 4, 32000, generator=g)
    mixture = sources.sum(1, keepdim=True)
    pp = {k: v.double().requires_grad_(True) for k, v in p.items()}
    ref_out, _ = FP.conv_tasnet(mixture.double(), pp, cfg)
    ref_loss, ref_P = O.sinkpit(lambda a, b, batch_mean=False: O.neg_sisdr(a, b, batch_mean=batch_mean), ref_out, sources.double(), coldness=1.0, iteration=200)
    ref_loss.backward()
    ref_grads = {k: v.grad for k, v in pp.items()}
    model.cuda()
    est = model(mixture.cuda())
    assert _rel(est, ref_out.detach()) <= TOL
    loss, pattern = SinkPIT(NegSISDR(), n_sources=4, coldness=1.0, iteration=200)(est, sources.cuda())
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    assert torch.equal(pattern.cpu().view(-1), ref_P.detach().argmax(dim=2).view(-1))
    loss.backward()
    flat_rel, worst = _grad_report(model, ref_grads)
    assert flat_rel <= TOL, "flat gradient rel-inf {:.3e}, worst tensor {}".format(flat_rel, worst)


@pytest.mark.parametrize("T", [80000, 64007, 8001])
def test_validation_regime_paper_best_one_utterance_any_length(T):
    """SURVEY.md section 8f rank 2 (reference egs/wsj0-mix/common/src/driver.py:166-206): validation runs ONE utterance of its natural
    length (up to 10 s @ 8 kHz) through the paper-best model under torch.no_grad().  Forward and PIT loss against the fp64 CPU port at
    10 s, at a length off every tile size (64007 -> 8000 frames + input padding) and at 1 s."""
    torch.manual_seed(31)
    model = ConvTasNet(**PAPER)
    p = {k: v.detach().clone().double() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(T)
    sources = 0.1 * torch.randn(1, 2, T, generator=g)
    mixture = sources.sum(1, keepdim=True)
    with torch.no_grad():
        ref_out, _ = FP.conv_tasnet(mixture.double(), p, PAPER)
        ref_loss, ref_pat = FP.neg_sisdr_pit(ref_out, sources.double())
        model.cuda().eval()
        est = model(mixture.cuda())
        loss, pattern = PIT1d(NegSISDR(), n_sources=2)(est, sources.cuda())
    assert est.shape == (1, 2, T)
    assert _rel(est, ref_out) <= TOL
    assert abs(loss.item() - ref_loss.item()) <= TOL * abs(ref_loss.item())
    assert torch.equal(pattern.cpu(), ref_pat)


def test_music_recipe_shapes_long_rows_and_generic_filterbank_paths():
    """SURVEY.md section 8f rank 3 (musdb18 / WHAM callers): stereo input, kernel 20 / stride 10 (Cout*L = 40: the generic
    encoder / decoder kernels), 2 s @ 44.1 kHz -> 8,819 frames per row (> 7,680: the tiled depthwise backward), 4 sources.
    Truth: the fp64 CPU port."""
    cfg = dict(n_basis=128, kernel_size=20, stride=10, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
               sep_hidden_channels=256, sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3, sep_num_blocks=2,
               sep_num_layers=4, dilated=True, separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True,
               mask_nonlinear="sigmoid", n_sources=4, in_channels=2)
    torch.manual_seed(21)
    model = ConvTasNet(**cfg)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    T = 88200
    sources = 0.1 * torch.randn(1, 4, 2, T, generator=g)
    mixture = sources.sum(1, keepdim=True)                               # (1, 1, 2, T): the reference's 4-D stereo form
    ref_out, _ = FP.conv_tasnet(mixture.double().reshape(1, 2, T), {k: v.double() for k, v in p.items()}, model.get_config())
    model.cuda()
    est = model(mixture.cuda())
    assert est.shape == (1, 4, 2, T)
    assert _rel(est.reshape(ref_out.shape), ref_out) <= TOL
    # backward runs through the long-row kernels; gradient checked by a directional finite difference of a quadratic loss
    tgt = sources.cuda()
    loss = ((est - tgt) ** 2).mean()
    loss.backward()
    params = list(model.parameters())
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in params)
    torch.manual_seed(4)
    dirs = [torch.randn_like(q) * q.detach().abs().mean().clamp_min(1e-3) for q in params]
    analytic = sum((q.grad.double() * dd.double()).sum().item() for q, dd in zip(params, dirs))
    h, vals = 2e-3, []
    with torch.no_grad():
        for sgn in (+1, -1):
            for q, dd in zip(params, dirs):
                q.add_(sgn * h * dd)
            vals.append(((model(mixture.cuda()) - tgt) ** 2).mean().double().item())
            for q, dd in zip(params, dirs):
                q.sub_(sgn * h * dd)
    numeric = (vals[0] - vals[1]) / (2 * h)
    assert abs(analytic - numeric) <= 3e-2 * abs(numeric) + 1e-6, (analytic, numeric)



def test_distance_and_sdr_criteria_on_gpu():
    """criterion/distance.py + criterion/sdr.py:{SDR,NegSDR} (the music recipe's --criterion mae|mse|sdr) against the
    reference formulas in fp64 torch: value and gradient 1e-5 relative."""
    from criterion.distance import L1Loss, L2Loss, MeanAbsoluteError, MeanSquaredError
    from criterion.sdr import NegSDR
    g = torch.Generator().manual_seed(31)
    t = 0.1 * torch.randn(2, 4, 2, 44100, generator=g)
    x = t + 0.05 * torch.randn(2, 4, 2, 44100, generator=g)

    def formulas(xd, td):
        d = xd - td
        sdr = 10 * torch.log10(((td ** 2).sum(-1) + 1e-12) / ((d ** 2).sum(-1) + 1e-12))
        return {"mae": d.abs().mean(-1).mean((1, 2)).mean(0), "mse": (d ** 2).mean(-1).mean((1, 2)).mean(0),
                "l1": d.abs().sum(2).mean((1, 2)).mean(0), "l2": torch.sqrt((d ** 2).sum((2, 3))).sum(1).mean(0),
                "negsdr": -sdr.mean((1, 2)).mean(0)}

    crits = {"mae": MeanAbsoluteError(dim=-1, reduction="mean"), "mse": MeanSquaredError(dim=-1, reduction="mean"),
             "l1": L1Loss(dim=2), "l2": L2Loss(dim=(2, 3), reduction="sum"), "negsdr": NegSDR()}
    for name, crit in crits.items():
        xd = x.double().requires_grad_(True)
        want = formulas(xd, t.double())[name]
        want.backward()
        xg = x.cuda().requires_grad_(True)
        got = crit(xg, t.cuda())
        got.backward()
        assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item()), name
        assert _rel(xg.grad, xd.grad) <= 1e-5, name


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_single_rank_runs_the_exchange_path_of_the_train_step():
    """round-4 verdict item 4: the `nccl` backend (= RCCL on ROCm) for real on the one GPU a test box has.  A process group of ONE rank makes
    every collective the identity, so FusedTrainStep(exercise_collectives=True) -- broadcast of the flat parameter buffer, one asynchronous
    all_reduce per TCN block on slices of the flat gradient buffer issued from inside backward, the waits and their event bracket --
    must leave the parameters where the non-distributed step leaves them (to the run-to-run noise of the fp64 statistics atomics), with the weight gradients on and off their side stream."""
    import torch.distributed as dist
    from sepkernels.train import FusedTrainStep
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{}".format(_free_port()), rank=0, world_size=1)
    prev = os.environ.get("SEPK_SIDE_STREAM")
    try:
        cfg = dict(CONFIGS["mid"], n_sources=2)                  # two TCN blocks -> two gradient buckets
        crit = PIT1d(NegSISDR(), n_sources=2)
        g = torch.Generator().manual_seed(7)
        batches = [0.1 * torch.randn(3, 2, 3203, generator=g) for _ in range(3)]
        for side in ("0", "1"):
            os.environ["SEPK_SIDE_STREAM"] = side
            torch.manual_seed(11)
            a = ConvTasNet(**cfg).cuda()
            b = ConvTasNet(**cfg).cuda()
            b.load_state_dict(a.state_dict())
            plain = FusedTrainStep(a, crit, lr=1e-3, max_norm=5.0, distributed=False)
            comm = FusedTrainStep(b, crit, lr=1e-3, max_norm=5.0, time_collectives=True, exercise_collectives=True)
            assert comm.distributed and comm.world == 1 and comm.comm and not plain.comm
            rec = None
            if side == "0":     # round 6: the RECORDED step under the exchange path -- segments of the launch list between the buckets' all-reduces
                c = ConvTasNet(**cfg).cuda()
                c.load_state_dict(a.state_dict())
                rec = FusedTrainStep(c, crit, lr=1e-3, max_norm=5.0, time_collectives=True, exercise_collectives=True, auto_record=True)
                assert rec.comm and rec.recordable() is None
            for src in batches:
                src = src.cuda()
                mix = src.sum(1, keepdim=True).contiguous()
                la, lb = plain(mix, src), comm(mix, src)
                assert abs(la.item() - lb.item()) <= 1e-5 * abs(la.item())      # (fp64 atomics of the statistics: the last bits may differ run to run)
                if rec is not None:
                    lc = rec(mix, src)
                    assert abs(la.item() - lc.item()) <= 1e-5 * abs(la.item())
            if rec is not None:
                assert rec._seq is not None and len(rec._seq_marks) == cfg["sep_num_blocks"] and rec.last_buckets == cfg["sep_num_blocks"]
                assert sum(rec.last_bucket_bytes) == 4 * rec.gflat.numel() and rec.step_count == len(batches)
                ms = rec.exposed_comm_ms()
                assert ms is not None and np.isfinite(ms) and ms >= 0.0
                assert (a.flat_parameters() - c.flat_parameters()).abs().max().item() <= 1e-5 * a.flat_parameters().abs().max().item()
            assert comm.last_buckets == cfg["sep_num_blocks"] and sum(comm.last_bucket_bytes) == 4 * comm.gflat.numel()
            ms = comm.exposed_comm_ms()
            assert ms is not None and np.isfinite(ms) and ms >= 0.0
            assert (a.flat_parameters() - b.flat_parameters()).abs().max().item() <= 1e-5 * a.flat_parameters().abs().max().item(), "side stream " + side
            # the single-call path (SEPK_DDP_BUCKETS=0) as well
            os.environ["SEPK_DDP_BUCKETS"] = "0"
            try:
                one = FusedTrainStep(b, crit, lr=1e-3, max_norm=5.0, time_collectives=True, exercise_collectives=True)
                one.m, one.v, one.step_count = comm.m.clone(), comm.v.clone(), comm.step_count
            finally:
                del os.environ["SEPK_DDP_BUCKETS"]
            src = batches[0].cuda()
            mix = src.sum(1, keepdim=True).contiguous()
            plain(mix, src), one(mix, src)
            assert one.last_buckets == 0 and one.last_bucket_bytes == [4 * one.gflat.numel()]
            assert (a.flat_parameters() - b.flat_parameters()).abs().max().item() <= 1e-5 * a.flat_parameters().abs().max().item()
    finally:
        if prev is None:
            os.environ.pop("SEPK_SIDE_STREAM", None)
        else:
            os.environ["SEPK_SIDE_STREAM"] = prev
        dist.destroy_process_group()


def test_side_stream_on_and_off_give_the_same_gradients():
    """the weight gradients on their own stream (SEPK_SIDE_STREAM=1; off by default since round 5) and on the main stream are the same kernels in the
    same per-tensor order: the same gradients (to the run-to-run noise of the statistics' fp64 atomics) at a size where the streams really overlap"""
    cfg = dict(PAPER, sep_num_blocks=1, sep_num_layers=4)
    torch.manual_seed(3)
    model = ConvTasNet(**cfg).cuda()
    src = (0.1 * torch.randn(4, 2, 16000)).cuda()
    mix = src.sum(1, keepdim=True).contiguous()
    crit = PIT1d(NegSISDR(), n_sources=2)
    prev = os.environ.get("SEPK_SIDE_STREAM")
    grads = {}
    try:
        for side in ("0", "1"):
            os.environ["SEPK_SIDE_STREAM"] = side
            model.zero_grad(set_to_none=True)
            loss, _ = crit(model(mix), src)
            loss.backward()
            torch.cuda.synchronize()
            grads[side] = {k: p.grad.clone() for k, p in model.named_parameters()}
    finally:
        if prev is None:
            os.environ.pop("SEPK_SIDE_STREAM", None)
        else:
            os.environ["SEPK_SIDE_STREAM"] = prev
    for k in grads["0"]:
        assert (grads["0"][k] - grads["1"][k]).abs().max().item() <= 1e-5 * grads["0"][k].abs().max().item() + 1e-12, k
