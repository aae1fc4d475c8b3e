"""
ORACLE TOOLING -- TEST INFRASTRUCTURE ONLY.

Generates tests/golden/*.npz by importing the UNMODIFIED reference from
/root/reference/src (build container only; /root/reference does not exist on
the GPU box, which is why the vectors are committed).  Run:

    python oracle/make_golden.py

`torchaudio` is absent from the image; the reference imports it transitively
(models/filterbank.py:8 -> utils/audio.py:5) without using it on this path, so
an empty stub module is injected before the import.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def import_reference():
    sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.conv_tasnet import ConvTasNet
    from criterion.sdr import NegSISDR, SISDR
    from criterion.pit import PIT1d, SinkPIT
    return ConvTasNet, NegSISDR, SISDR, PIT1d, SinkPIT


CONFIGS = {
    # BASELINE.json configs[0]: tiny, enc ReLU, 2 speakers
    "tiny": dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
                 sep_hidden_channels=128, sep_bottleneck_channels=64, sep_skip_channels=64, sep_kernel_size=3,
                 sep_num_blocks=1, sep_num_layers=2, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
                 sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    # multi-block / dual-head logic, linear encoder, 3 speakers, skip != bottleneck width
    "mid": dict(n_basis=128, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                sep_hidden_channels=256, sep_bottleneck_channels=128, sep_skip_channels=64, sep_kernel_size=3,
                sep_num_blocks=2, sep_num_layers=3, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
                sep_norm=True, mask_nonlinear="sigmoid", n_sources=3),
    # mask_nonlinear='softmax' (nn.Softmax(dim=1) over the n_src*N channels of a frame, reference conv_tasnet.py:357)
    "softmax": dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
                    sep_hidden_channels=128, sep_bottleneck_channels=64, sep_skip_channels=64, sep_kernel_size=3,
                    sep_num_blocks=1, sep_num_layers=2, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
                    sep_norm=True, mask_nonlinear="softmax", n_sources=2),
    # --- configurations OUTSIDE the fused kernel family: the product runs them as the module-by-module composition
    # the reference constructor's default: causal = True -> cLN everywhere, all padding on the left
    "causal": dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
                   sep_hidden_channels=96, sep_bottleneck_channels=48, sep_skip_channels=40, sep_kernel_size=3,
                   sep_num_blocks=2, sep_num_layers=3, dilated=True, separable=True, causal=True, sep_nonlinear="prelu",
                   sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    # the causal family on channel counts the kernels take (multiples of 16): runs layer by layer on this library's kernels ("staged":
    # models/conv_tasnet.py::_run_staged), softmax mask over 3 speakers, linear encoder, 20 / 10 basis
    "causal16": dict(n_basis=64, kernel_size=20, stride=10, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                     sep_hidden_channels=96, sep_bottleneck_channels=48, sep_skip_channels=32, sep_kernel_size=3,
                     sep_num_blocks=2, sep_num_layers=3, dilated=True, separable=True, causal=True, sep_nonlinear="prelu",
                     sep_norm=True, mask_nonlinear="softmax", n_sources=3),
    # the reference constructor's defaults where it has them (causal=True, sigmoid, 2 speakers), enc ReLU, 5 taps
    "causal16_p5": dict(n_basis=32, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
                        sep_hidden_channels=64, sep_bottleneck_channels=32, sep_skip_channels=48, sep_kernel_size=5,
                        sep_num_blocks=1, sep_num_layers=4, dilated=True, separable=True, causal=True, sep_nonlinear="prelu",
                        sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    # a 128-row bottleneck: the two heads of a layer run as ONE product over [Wo; Ws] (sepkernels.functional.PaddedHeadsFn)
    "causal16_joint": dict(n_basis=32, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                           sep_hidden_channels=64, sep_bottleneck_channels=128, sep_skip_channels=32, sep_kernel_size=3,
                           sep_num_blocks=2, sep_num_layers=2, dilated=True, separable=True, causal=True, sep_nonlinear="prelu",
                           sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    # full (non-separable) 5-tap convolutions, channel counts off the multiples of 16
    "plainconv": dict(n_basis=60, kernel_size=20, stride=10, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                      sep_hidden_channels=72, sep_bottleneck_channels=36, sep_skip_channels=28, sep_kernel_size=5,
                      sep_num_blocks=2, sep_num_layers=2, dilated=True, separable=False, causal=False, sep_nonlinear="prelu",
                      sep_norm=True, mask_nonlinear="sigmoid", n_sources=3),
    # not dilated (stride-2 geometry of the padding formula), no norm, no activation inside the TCN, softmax mask
    "nodil": dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
                  sep_hidden_channels=64, sep_bottleneck_channels=32, sep_skip_channels=32, sep_kernel_size=3,
                  sep_num_blocks=1, sep_num_layers=3, dilated=False, separable=True, causal=False, sep_nonlinear=None,
                  sep_norm=False, mask_nonlinear="softmax", n_sources=2),
    # --- other filterbanks (reference models/filterbank.py:12-203, 253-323): fixed Fourier basis with complex latent (the magnitude is
    # masked, the phase kept), trainable frequencies + phases with a real-valued one-sided latent, pseudo-inverse decoders
    "fourier": dict(n_basis=17, kernel_size=32, stride=8, enc_basis="Fourier", dec_basis="Fourier", enc_nonlinear=None, window_fn="hann",
                    enc_onesided=1, enc_return_complex=1, sep_hidden_channels=48, sep_bottleneck_channels=32, sep_skip_channels=32,
                    sep_kernel_size=3, sep_num_blocks=1, sep_num_layers=2, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
                    sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    "fourier_phase": dict(n_basis=34, kernel_size=32, stride=16, enc_basis="trainableFourierTrainablePhase", dec_basis="trainableFourierTrainablePhase",
                          enc_nonlinear=None, window_fn="hamming", enc_onesided=1, enc_return_complex=0, sep_hidden_channels=48,
                          sep_bottleneck_channels=32, sep_skip_channels=32, sep_kernel_size=3, sep_num_blocks=1, sep_num_layers=2, dilated=True,
                          separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    "fourier_pinv": dict(n_basis=32, kernel_size=16, stride=8, enc_basis="trainableFourier", dec_basis="pinv", enc_nonlinear=None, window_fn="hann",
                         enc_onesided=0, enc_return_complex=0, sep_hidden_channels=48, sep_bottleneck_channels=32, sep_skip_channels=32,
                         sep_kernel_size=3, sep_num_blocks=1, sep_num_layers=2, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
                         sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    # the reference recipe's settings for its published "Fourier / Fourier" row (train.sh:21-26: two-sided, real-valued latent) and the same
    # with trainable frequencies and phases on both sides: linear filterbanks -> the fused kernel sequence on derived bases
    "fourier_real": dict(n_basis=64, kernel_size=16, stride=8, enc_basis="Fourier", dec_basis="Fourier", enc_nonlinear=None, window_fn="hann",
                         enc_onesided=0, enc_return_complex=0, sep_hidden_channels=64, sep_bottleneck_channels=32, sep_skip_channels=32,
                         sep_kernel_size=3, sep_num_blocks=1, sep_num_layers=3, dilated=True, separable=True, causal=False, sep_nonlinear="prelu",
                         sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
    "fourier_phase_real": dict(n_basis=32, kernel_size=32, stride=16, enc_basis="trainableFourierTrainablePhase", dec_basis="trainableFourierTrainablePhase",
                               enc_nonlinear=None, window_fn="hamming", enc_onesided=0, enc_return_complex=0, sep_hidden_channels=48,
                               sep_bottleneck_channels=32, sep_skip_channels=32, sep_kernel_size=3, sep_num_blocks=1, sep_num_layers=2, dilated=True,
                               separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear="softmax", n_sources=3),
    "pinv": dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="pinv", enc_nonlinear=None, sep_hidden_channels=48,
                 sep_bottleneck_channels=32, sep_skip_channels=32, sep_kernel_size=3, sep_num_blocks=1, sep_num_layers=2, dilated=True,
                 separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True, mask_nonlinear="sigmoid", n_sources=2),
}
SHAPES = {"tiny": (1, 4000), "mid": (2, 3203), "softmax": (2, 2500),   # (batch, samples); 3203 / 2500 exercise the input padding branch
          "causal": (2, 2403), "causal16": (2, 2403), "causal16_p5": (3, 1500), "causal16_joint": (2, 1203), "plainconv": (2, 2000), "nodil": (1, 1607),
          "fourier": (2, 1603), "fourier_phase": (2, 1603), "fourier_pinv": (1, 1603), "pinv": (2, 1603), "fourier_real": (2, 1603),
          "fourier_phase_real": (2, 1603)}
COMPOSED = ("causal", "plainconv", "nodil", "fourier", "fourier_phase", "fourier_pinv", "pinv", "fourier_real", "fourier_phase_real")
DERIVED = ("fourier_pinv", "pinv", "fourier_real", "fourier_phase_real")      # linear filterbanks: the fused sequence on bases formed from their parameters
STAGED = ("causal16", "causal16_p5", "causal16_joint")      # causal, on kernels layer by layer


def perturb(model, seed):
    """Default init has gamma=1, beta=0, alpha=0.25, which hides gamma/beta/alpha
    indexing mistakes.  Perturb them (deterministically) so every parameter matters."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("norm.weight"):
                p.add_(0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("norm.bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("nonlinear1d.weight") or name.endswith("prelu.weight"):
                p.add_(0.1 * torch.rand(p.shape, generator=g))


def model_golden(name, ConvTasNet, NegSISDR, PIT1d):
    cfg = CONFIGS[name]
    B, T = SHAPES[name]
    torch.manual_seed(111)
    model = ConvTasNet(**cfg)
    perturb(model, 7)
    g = torch.Generator().manual_seed(222)
    sources = 0.1 * torch.randn(B, cfg["n_sources"], T, generator=g)
    mixture = sources.sum(dim=1, keepdim=True)
    crit = PIT1d(NegSISDR(), n_sources=cfg["n_sources"])

    # fp32: the reference exactly as shipped
    out32, latent32 = model.extract_latent(mixture)
    loss32, pattern = crit(out32, sources)
    # fp64 run of the same module tree: ground truth for gradients (SURVEY 8c noise-floor note)
    import copy
    m64 = copy.deepcopy(model).double()
    out64, latent64 = m64.extract_latent(mixture.double())
    loss64, pattern64 = crit(out64, sources.double())
    loss64.backward()
    assert torch.equal(pattern, pattern64)

    blob = {"mixture": mixture.numpy(), "sources": sources.numpy(),
            "output_f32": out32.detach().numpy(), "output_f64": out64.detach().numpy(),
            "latent_f64_sum": np.array(latent64.detach().sum().real.item() if torch.is_complex(latent64) else latent64.detach().sum().item()),
            "latent_f64_abs_sum": np.array(latent64.detach().abs().sum().item()),
            "loss_f32": np.array(loss32.item()), "loss_f64": np.array(loss64.item()),
            "pattern": pattern.numpy()}
    for k, v in model.state_dict().items():
        blob["param/" + k] = v.numpy()
    for k, p in m64.named_parameters():
        if p.grad is not None:             # (Fourier bases carry non-trainable parameters: time_seq, and frequency when not trainable)
            blob["grad/" + k] = p.grad.numpy().astype(np.float32)  # fp64 truth, stored as f32 to keep the fixture small
    blob["num_parameters"] = np.array(model.num_parameters)
    np.savez_compressed(os.path.join(OUT, "convtasnet_{}.npz".format(name)), **blob)
    print(name, "params", model.num_parameters, "loss", loss64.item(), "pattern", pattern.tolist())


def pit_kat(NegSISDR, SISDR, PIT1d, SinkPIT):
    """The reference's own deterministic self-test, criterion/pit.py:226-263 and :331-361."""
    import random
    blob = {}
    torch.manual_seed(111)
    x = torch.randint(2, (4, 2, 1024), dtype=torch.float)
    t = torch.randint(2, (4, 2, 1024), dtype=torch.float)
    loss, pattern = PIT1d(SISDR(), n_sources=2)(x, t)
    blob.update(pit_x=x.numpy(), pit_t=t.numpy(), pit_sisdr_loss=np.array(loss.item()), pit_sisdr_pattern=pattern.numpy())
    print("PIT SI-SDR", loss.item(), pattern.tolist())

    random.seed(111)
    torch.manual_seed(111)
    x = torch.randint(2, (4, 3, 1024), dtype=torch.float)
    t = torch.randint(2, (4, 3, 1024), dtype=torch.float)
    loss, pattern = PIT1d(NegSISDR(), n_sources=3)(x, t)
    blob.update(sink_x=x.numpy(), sink_t=t.numpy(), pit3_negsisdr_loss=np.array(loss.item()), pit3_negsisdr_pattern=pattern.numpy())
    print("PIT NegSI-SDR", loss.item(), pattern.tolist())
    loss, pattern = SinkPIT(NegSISDR(), n_sources=3, coldness=1)(x, t, batch_mean=False)
    blob.update(sinkpit_neg_loss=loss.numpy(), sinkpit_neg_pattern=pattern.numpy())
    print("SinkPIT NegSI-SDR", loss.tolist(), pattern.tolist())
    loss, pattern = SinkPIT(SISDR(), n_sources=3, coldness=1)(x, t, batch_mean=False)
    blob.update(sinkpit_pos_loss=loss.numpy(), sinkpit_pos_pattern=pattern.numpy())
    print("SinkPIT SI-SDR", loss.tolist(), pattern.tolist())

    # gradient of SinkPIT through all iterations (fp64), continuous inputs, 4 sources, k=20, beta=2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 4, 777, generator=g, dtype=torch.float64).requires_grad_(True)
    t = torch.randn(3, 4, 777, generator=g, dtype=torch.float64)
    x.data += 0.7 * t[:, [2, 0, 3, 1]]
    loss, pattern = SinkPIT(NegSISDR(), n_sources=4, coldness=2.0, iteration=20)(x, t)
    loss.backward()
    blob.update(sg_x=x.detach().numpy(), sg_t=t.numpy(), sg_loss=np.array(loss.item()), sg_pattern=pattern.numpy(),
                sg_grad=x.grad.numpy())
    # ORPIT (pit.py:87-161) with a variable number of sources per item (packed sequence), fp64
    from criterion.pit import ORPIT
    import torch.nn as nn
    g = torch.Generator().manual_seed(17)
    lens = [3, 2, 4]
    tg = [torch.randn(n, 500, generator=g, dtype=torch.float64) for n in lens]
    x = torch.stack([torch.stack([t[1 % len(t)] + 0.3 * torch.randn(500, generator=g, dtype=torch.float64),
                                  t.sum(0) - t[1 % len(t)] + 0.3 * torch.randn(500, generator=g, dtype=torch.float64)]) for t in tg])
    x.requires_grad_(True)
    packed = nn.utils.rnn.pack_sequence(tg, enforce_sorted=False)
    loss, idx = ORPIT(NegSISDR())(x, packed, batch_mean=False)
    loss.sum().backward()
    blob.update(orpit_x=x.detach().numpy(), orpit_loss=loss.detach().numpy(), orpit_idx=idx.numpy(), orpit_grad=x.grad.numpy(),
                orpit_lens=np.array(lens), orpit_t=nn.utils.rnn.pad_sequence(tg, batch_first=True).numpy())
    print("ORPIT", loss.tolist(), idx.tolist())
    np.savez_compressed(os.path.join(OUT, "pit_kat.npz"), **blob)


def op_golden(NegSISDR):
    """Module-level vectors for the building blocks (reference modules run directly)."""
    from modules.norm import GlobalLayerNorm
    from models.tdcn import TimeDilatedConvNet
    from models.filterbank import Encoder, Decoder
    blob = {}
    torch.manual_seed(3)
    x = torch.randn(2, 6, 50, dtype=torch.float64) * 2 + 0.5
    norm = GlobalLayerNorm(6).double()
    with torch.no_grad():
        norm.norm.weight.copy_(torch.randn(6)); norm.norm.bias.copy_(torch.randn(6))
    blob.update(gln_x=x.numpy(), gln_w=norm.norm.weight.detach().numpy(), gln_b=norm.norm.bias.detach().numpy(),
                gln_y=norm(x).detach().numpy())
    net = TimeDilatedConvNet(8, hidden_channels=12, skip_channels=10, kernel_size=3, num_blocks=2, num_layers=3,
                             dilated=True, separable=True, causal=False, nonlinear="prelu", norm=True).double()
    x = torch.randn(2, 8, 37, dtype=torch.float64)
    blob.update(tdcn_x=x.numpy(), tdcn_y=net(x).detach().numpy())
    for k, v in net.state_dict().items():
        blob["tdcn/" + k] = v.numpy()
    enc = Encoder(2, 7, kernel_size=6, stride=3, nonlinear="relu").double()
    x = torch.randn(2, 2, 33, dtype=torch.float64)
    blob.update(enc_x=x.numpy(), enc_w=enc.conv1d.weight.detach().numpy(), enc_y=enc(x).detach().numpy())
    dec = Decoder(7, 2, kernel_size=6, stride=3).double()
    w = torch.randn(3, 7, 10, dtype=torch.float64)
    blob.update(dec_x=w.numpy(), dec_w=dec.conv_transpose1d.weight.detach().numpy(), dec_y=dec(w).detach().numpy())
    a = torch.randn(3, 2, 100, dtype=torch.float64)
    b = torch.randn(3, 2, 100, dtype=torch.float64)
    blob.update(sdr_x=a.numpy(), sdr_t=b.numpy(), sdr_negsisdr=NegSISDR()(a, b, batch_mean=False).numpy())
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **blob)


DPRNN_CFG = dict(n_basis=64, kernel_size=2, stride=1, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                 sep_hidden_channels=32, sep_bottleneck_channels=64, sep_chunk_size=20, sep_hop_size=10, sep_num_blocks=2,
                 sep_norm=True, mask_nonlinear="sigmoid", causal=False, rnn_type="lstm", n_sources=2)


def dprnn_golden(NegSISDR, PIT1d):
    """DPRNN-TasNet (reference src/models/dprnn_tasnet.py), small configuration, fp64 module run as ground truth."""
    import copy
    from models.dprnn_tasnet import DPRNNTasNet
    torch.manual_seed(111)
    model = DPRNNTasNet(**DPRNN_CFG)
    perturb(model, 11)
    g = torch.Generator().manual_seed(333)
    sources = 0.1 * torch.randn(2, 2, 813, generator=g)          # 812 frames -> exercises the chunk padding
    mixture = sources.sum(dim=1, keepdim=True)
    m64 = copy.deepcopy(model).double()
    out64, latent64 = m64.extract_latent(mixture.double())
    loss64, pattern = PIT1d(NegSISDR(), n_sources=2)(out64, sources.double())
    loss64.backward()
    blob = {"mixture": mixture.numpy(), "sources": sources.numpy(), "output_f64": out64.detach().numpy(),
            "latent_f64_sum": np.array(latent64.detach().sum().real.item() if torch.is_complex(latent64) else latent64.detach().sum().item()), "loss_f64": np.array(loss64.item()),
            "pattern": pattern.numpy(), "num_parameters": np.array(model.num_parameters)}
    for k, v in model.state_dict().items():
        blob["param/" + k] = v.numpy()
    for k, p in m64.named_parameters():
        blob["grad/" + k] = p.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "dprnn_tasnet_small.npz"), **blob)
    print("dprnn params", model.num_parameters, "loss", loss64.item(), "pattern", pattern.tolist())


# BASELINE.json configs[3] at its REAL size (reference egs/wsj0-mix/dprnn-tasnet/train.sh:28-37), one utterance of 4 s @ 8 kHz
DPRNN_FULL_CFG = dict(n_basis=64, kernel_size=2, stride=1, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                      sep_hidden_channels=128, sep_bottleneck_channels=64, sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6,
                      sep_norm=True, mask_nonlinear="sigmoid", causal=False, rnn_type="lstm", n_sources=2)
DPRNN_FULL_SEEDS = dict(model=111, perturb=11, data=333)


def sample_index(n, k=64):
    """the k positions of a flat tensor of n elements a fingerprint keeps (first, last and an even spread)"""
    return sorted(set(int(round(q * (n - 1) / max(k - 1, 1))) for q in range(min(k, n))))


def dprnn_full_golden(NegSISDR, PIT1d):
    """The full-size DPRNN-TasNet of the REFERENCE on one seeded utterance, fp64: output, PIT loss, permutation and a fingerprint of every
    parameter gradient (L2 norm, largest magnitude, 64 sampled elements).  The 2.6 M parameters are not stored: default initialisation under
    the seed (the product's classes draw the same values, tests check the parameter fingerprints first) + perturb()."""
    import copy
    from models.dprnn_tasnet import DPRNNTasNet
    torch.manual_seed(DPRNN_FULL_SEEDS["model"])
    model = DPRNNTasNet(**DPRNN_FULL_CFG)
    perturb(model, DPRNN_FULL_SEEDS["perturb"])
    g = torch.Generator().manual_seed(DPRNN_FULL_SEEDS["data"])
    sources = 0.1 * torch.randn(1, 2, 32000, generator=g)
    mixture = sources.sum(dim=1, keepdim=True)
    m64 = copy.deepcopy(model).double()
    out64 = m64(mixture.double())
    loss64, pattern = PIT1d(NegSISDR(), n_sources=2)(out64, sources.double())
    loss64.backward()
    blob = {"output_f64": out64.detach().numpy().astype(np.float32), "loss_f64": np.array(loss64.item()), "pattern": pattern.numpy(),
            "num_parameters": np.array(model.num_parameters), "mixture_head": mixture.numpy()[0, 0, :16]}
    for k, v in model.state_dict().items():
        blob["pfp/" + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    for k, p in m64.named_parameters():
        gr = p.grad.reshape(-1)
        blob["gfp/" + k] = np.concatenate([[gr.norm().item(), gr.abs().max().item()], gr[sample_index(gr.numel())].numpy()])
    np.savez_compressed(os.path.join(OUT, "dprnn_tasnet_full.npz"), **blob)
    print("dprnn full: params", model.num_parameters, "loss", loss64.item(), "pattern", pattern.tolist())



# BASELINE.json configs[1] at its real batch (paper-best Conv-TasNet, 16 utterances of 4 s): the REFERENCE in fp64, once, here -- the GPU
# box has no reference and its fp64 port of this batch takes minutes.  What is stored: per-utterance output fingerprints (norm, largest
# magnitude, 64 samples per source), per-utterance and batch PIT loss, permutations, and a fingerprint of every parameter gradient (norm,
# largest magnitude, 64 samples; scalars whole).  The 5 M parameters are not stored: default init under the seed + the perturbation below
# (the product's class draws the same values; tests check the parameter fingerprints first).
PAPER_CFG = dict(n_basis=512, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                 sep_hidden_channels=512, sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3, sep_num_blocks=3,
                 sep_num_layers=8, dilated=True, separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True,
                 mask_nonlinear="sigmoid", n_sources=2)
PAPER_B16_SEEDS = {"model": 111, "data": 5}


def paper_b16_inputs(model):
    """the batch and the parameter perturbation of tests/test_gpu_model.py::test_batch16_* (one generator, in this order)"""
    g = torch.Generator().manual_seed(PAPER_B16_SEEDS["data"])
    with torch.no_grad():
        for n, q in model.named_parameters():
            if n.endswith("norm.weight") or n.endswith("norm.bias"):
                q.add_(0.1 * torch.randn(q.shape, generator=g))
    sources = 0.1 * torch.randn(16, 2, 32000, generator=g) * torch.exp(0.7 * torch.randn(16, 2, 1, generator=g))     # utterances / speakers of different level
    return sources.sum(1, keepdim=True), sources


def paper_b16_golden(ConvTasNet, NegSISDR, PIT1d):
    import time
    torch.manual_seed(PAPER_B16_SEEDS["model"])
    model = ConvTasNet(**PAPER_CFG)
    mixture, sources = paper_b16_inputs(model)
    blob = {"mixture_head": mixture.numpy()[:, 0, :8].copy()}
    for k, v in model.state_dict().items():
        blob["pfp/" + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    t0 = time.time()
    m64 = model.double()
    out64 = m64(mixture.double())
    crit = PIT1d(NegSISDR(), n_sources=2)
    loss64, pattern = crit(out64, sources.double())
    per_utt, _ = crit(out64.detach(), sources.double(), batch_mean=False)
    loss64.backward()
    o = out64.detach()
    idx = sample_index(32000)
    blob.update({"loss_f64": np.array(loss64.item()), "per_utt_f64": per_utt.numpy(), "pattern": pattern.numpy(),
                 "out_norm": o.norm(dim=2).numpy(), "out_amax": o.abs().amax(dim=2).numpy(), "out_idx": np.array(idx), "out_samples": o[:, :, idx].numpy()})
    for k, p in m64.named_parameters():
        gr = p.grad.reshape(-1)
        blob["gfp/" + k] = np.concatenate([[gr.norm().item(), gr.abs().max().item()], gr[sample_index(gr.numel())].numpy()])
    np.savez_compressed(os.path.join(OUT, "convtasnet_paper_b16.npz"), **blob)
    print("paper-best B=16 fp64 reference: loss", loss64.item(), "in {:.0f} s".format(time.time() - t0))


# The reference's TRAINING STEP, several times over (egs/wsj0-mix/common/src/driver.py:141-157: zero_grad, model(mixture), pit_criterion,
# backward, clip_grad_norm_(max_norm), optimizer.step with torch.optim.Adam -- local/train.py:103-118), on seeded batches, in fp32 as shipped
# and in fp64: the loss of every step and a fingerprint of the parameters afterwards.  The GPU tier runs the recipe's own step
# (sepkernels.train.FusedTrainStep, what recipes.trainer.Trainer drives) on the same batches and has to stay on this trajectory.
TRAJ_CFG = dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu",
                sep_hidden_channels=128, sep_bottleneck_channels=64, sep_skip_channels=64, sep_kernel_size=3, sep_num_blocks=2,
                sep_num_layers=3, dilated=True, separable=True, causal=False, sep_nonlinear="prelu", sep_norm=True,
                mask_nonlinear="sigmoid", n_sources=2)
TRAJ = dict(model_seed=31, data_seed=32, steps=8, batch=4, samples=4000, lr=1e-3, max_norm=5.0)


def traj_batches():
    g = torch.Generator().manual_seed(TRAJ["data_seed"])
    out = []
    for _ in range(TRAJ["steps"]):
        sources = 0.1 * torch.randn(TRAJ["batch"], 2, TRAJ["samples"], generator=g) * torch.exp(0.5 * torch.randn(TRAJ["batch"], 2, 1, generator=g))
        out.append((sources.sum(1, keepdim=True), sources))
    return out


def train_trajectory_golden(ConvTasNet, NegSISDR, PIT1d):
    blob = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.manual_seed(TRAJ["model_seed"])
        model = ConvTasNet(**TRAJ_CFG).to(dt)
        crit = PIT1d(NegSISDR(), n_sources=2)
        opt = torch.optim.Adam(model.parameters(), lr=TRAJ["lr"])
        losses = []
        for mixture, sources in traj_batches():
            opt.zero_grad()
            loss, _ = crit(model(mixture.to(dt)), sources.to(dt))
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), TRAJ["max_norm"])
            opt.step()
            losses.append(loss.item())
        blob["loss_" + name] = np.array(losses)
        if name == "f64":
            for k, v in model.state_dict().items():
                blob["pfp/" + k] = np.array([v.double().sum().item(), v.double().abs().sum().item(), v.double().abs().max().item()])
    np.savez_compressed(os.path.join(OUT, "train_trajectory.npz"), **blob)
    print("training trajectory (reference step x {}): fp64 losses".format(TRAJ["steps"]), [round(v, 5) for v in blob["loss_f64"]], "fp32 - fp64",
          float(np.abs(blob["loss_f32"] - blob["loss_f64"]).max()))

# DPTNet / GALRNet / SepFormer: small configurations, channel counts in multiples of 16 so that the
# product runs them on its kernel path, plus one with odd widths (composition path).  403 samples -> 201 frames: both the
# waveform padding and the chunk padding (1 frame left, 2 right) are exercised.
_ENC = dict(kernel_size=4, stride=2, enc_basis="trainable", dec_basis="trainable")
SIBLINGS = {
    "dptnet": ("DPTNet", dict(n_basis=32, enc_nonlinear="relu", sep_bottleneck_channels=32, sep_hidden_channels=16, sep_chunk_size=12,
                              sep_num_blocks=2, sep_num_heads=4, sep_dropout=0, mask_nonlinear="relu", causal=False, n_sources=2, **_ENC)),
    "dptnet_causal": ("DPTNet", dict(n_basis=32, enc_nonlinear=None, sep_bottleneck_channels=16, sep_hidden_channels=32, sep_chunk_size=10,
                                     sep_hop_size=5, sep_num_blocks=1, sep_num_heads=2, sep_dropout=0, mask_nonlinear="sigmoid", causal=True,
                                     n_sources=3, **_ENC)),
    "dptnet_odd": ("DPTNet", dict(n_basis=24, enc_nonlinear="relu", sep_bottleneck_channels=20, sep_hidden_channels=12, sep_chunk_size=12,
                                  sep_num_blocks=1, sep_num_heads=4, sep_dropout=0, mask_nonlinear="softmax", causal=False, n_sources=2, **_ENC)),
    "galrnet": ("GALRNet", dict(n_basis=32, enc_nonlinear="relu", sep_hidden_channels=16, sep_chunk_size=12, sep_hop_size=6,
                                sep_down_chunk_size=4, sep_num_blocks=2, sep_num_heads=4, sep_dropout=0.0, mask_nonlinear="relu",
                                causal=False, n_sources=2, low_dimension=True, **_ENC)),
    "galrnet_causal": ("GALRNet", dict(n_basis=32, enc_nonlinear=None, sep_hidden_channels=16, sep_chunk_size=10, sep_hop_size=5,
                                       sep_num_blocks=1, sep_num_heads=2, sep_dropout=0.0, mask_nonlinear="sigmoid", causal=True,
                                       n_sources=2, low_dimension=False, **_ENC)),
    "sepformer": ("SepFormer", dict(n_basis=32, enc_nonlinear="relu", sep_bottleneck_channels=32, sep_chunk_size=12, sep_hop_size=6,
                                    sep_num_blocks=1, sep_num_layers_intra=2, sep_num_layers_inter=2, sep_num_heads_intra=4,
                                    sep_num_heads_inter=4, sep_d_ff_intra=48, sep_d_ff_inter=48, sep_dropout=0.0, mask_nonlinear="relu",
                                    causal=False, n_sources=2, **_ENC)),
    "sepformer_causal": ("SepFormer", dict(n_basis=32, enc_nonlinear=None, sep_bottleneck_channels=16, sep_chunk_size=10, sep_hop_size=5,
                                           sep_num_blocks=1, sep_num_layers_intra=1, sep_num_layers_inter=1, sep_num_heads_intra=2,
                                           sep_num_heads_inter=2, sep_d_ff_intra=24, sep_d_ff_inter=24, sep_dropout=0.0,
                                           mask_nonlinear="sigmoid", causal=True, n_sources=2, **_ENC)),
    # DPRNN-TasNet outside its head / tail kernel family (the small in-family instance is dprnn_tasnet_small above)
    "dprnn_tasnet_causal": ("DPRNNTasNet", dict(n_basis=32, enc_nonlinear=None, sep_hidden_channels=16, sep_bottleneck_channels=32,
                                                sep_chunk_size=10, sep_hop_size=5, sep_num_blocks=1, sep_norm=True, mask_nonlinear="sigmoid",
                                                causal=True, rnn_type="lstm", n_sources=2, **_ENC)),
    "dprnn_tasnet_odd": ("DPRNNTasNet", dict(n_basis=24, enc_nonlinear="relu", sep_hidden_channels=12, sep_bottleneck_channels=20,
                                             sep_chunk_size=12, sep_hop_size=6, sep_num_blocks=1, sep_norm=True, mask_nonlinear="softmax",
                                             causal=False, rnn_type="lstm", n_sources=2, **_ENC)),
    "dprnn_tasnet_softmax": ("DPRNNTasNet", dict(n_basis=32, enc_nonlinear="relu", sep_hidden_channels=16, sep_bottleneck_channels=32,
                                                 sep_chunk_size=12, sep_hop_size=6, sep_num_blocks=1, sep_norm=True, mask_nonlinear="softmax",
                                                 causal=False, rnn_type="lstm", n_sources=3, **_ENC)),
}
SIBLING_MODULES = {"DPTNet": "models.dptnet", "GALRNet": "models.galrnet", "SepFormer": "models.sepformer", "DPRNNTasNet": "models.dprnn_tasnet"}


def perturb_all(model, seed):
    """every normalisation gain / shift, every bias (attention and transformer biases start at zero) and the PReLU slope moved
    off their initial values, so that none of them can be dropped or mis-indexed unnoticed"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            leaf = name.split(".")[-1]
            if "norm" in name or leaf in ("gamma", "beta"):
                p.add_((0.2 if leaf in ("weight", "gamma") else 0.1) * torch.randn(p.shape, generator=g))
            elif leaf.startswith("bias") or leaf.endswith("bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            elif name.endswith("prelu.weight"):
                p.add_(0.1 * torch.rand(p.shape, generator=g))


def sibling_golden(name, NegSISDR, PIT1d):
    """reference src/models/{dptnet,galrnet,sepformer}.py, fp64 module run as ground truth"""
    import copy
    import importlib
    kind, cfg = SIBLINGS[name]
    cls = getattr(importlib.import_module(SIBLING_MODULES[kind]), kind)
    torch.manual_seed(111)
    model = cls(**cfg)
    perturb_all(model, 17)
    n_src = cfg["n_sources"]
    g = torch.Generator().manual_seed(444)
    sources = 0.1 * torch.randn(2, n_src, 403, generator=g)
    mixture = sources.sum(dim=1, keepdim=True)
    m64 = copy.deepcopy(model).double()
    out64, latent64 = m64.extract_latent(mixture.double())
    loss64, pattern = PIT1d(NegSISDR(), n_sources=n_src)(out64, sources.double())
    loss64.backward()
    blob = {"mixture": mixture.numpy(), "sources": sources.numpy(), "output_f64": out64.detach().numpy(),
            "latent_f64_sum": np.array(latent64.detach().sum().item()), "latent_f64_abs_sum": np.array(latent64.detach().abs().sum().item()),
            "loss_f64": np.array(loss64.item()), "pattern": pattern.numpy(), "num_parameters": np.array(model.num_parameters),
            "config_keys": np.array(list(model.get_config().keys()))}
    blob["state_keys"] = np.array(list(model.state_dict().keys()))
    for k, v in model.state_dict().items():
        if not k.endswith("positional_encoding.positional_encoding"):      # 5000 x C table of a closed formula per transformer stack: not stored
            blob["param/" + k] = v.numpy()
    for k, p in m64.named_parameters():
        blob["grad/" + k] = p.grad.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "{}.npz".format(name)), **blob)
    print(name, "params", model.num_parameters, "loss", loss64.item(), "pattern", pattern.tolist())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ConvTasNet, NegSISDR, SISDR, PIT1d, SinkPIT = import_reference()
    only = sys.argv[1:]                  # e.g. `python -m oracle.make_golden softmax` regenerates one model fixture
    for name in CONFIGS:
        if not only or name in only:
            model_golden(name, ConvTasNet, NegSISDR, PIT1d)
    for name in SIBLINGS:
        if not only or name in only:
            sibling_golden(name, NegSISDR, PIT1d)
    if not only:
        pit_kat(NegSISDR, SISDR, PIT1d, SinkPIT)
        op_golden(NegSISDR)
        dprnn_golden(NegSISDR, PIT1d)
    if not only or "dprnn_full" in only:
        dprnn_full_golden(NegSISDR, PIT1d)
    if not only or "paper_b16" in only:
        paper_b16_golden(ConvTasNet, NegSISDR, PIT1d)
    if not only or "trajectory" in only:
        train_trajectory_golden(ConvTasNet, NegSISDR, PIT1d)
    print("golden vectors written to", OUT)
