"""CPU: DPTNet, GALRNet and SepFormer -- and DPRNN-TasNet outside its head / tail kernel family (causal,
odd widths; softmax mask inside it) -- through the public module API with the CPU emulator of the C ABI,
against golden vectors generated from the unmodified reference (oracle/make_golden.py::sibling_golden, fp64 run): state_dict key
list and order, parameter count, config keys, forward, latent, PIT loss, permutation and EVERY parameter gradient -- on the
kernel path (channel counts in multiples of 16) and on the composition path (odd widths, softmax mask), both checked to be the
path actually taken.  Plus the two paths against each other and the pieces with no counterpart elsewhere (GTU, position codes)."""
import os

import numpy as np
import pytest
import torch

import sepkernels
from emulator import EmuBackend
from oracle.make_golden import SIBLINGS
from models.dptnet import DPTNet
from models.galrnet import GALRNet
from models.sepformer import SepFormer
from models.dprnn_tasnet import DPRNNTasNet
from models.gtu import GTU1d
from models.transformer import PositionalEncoding
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d

CLASSES = {"DPTNet": DPTNet, "GALRNet": GALRNet, "SepFormer": SepFormer, "DPRNNTasNet": DPRNNTasNet}
COMPOSED = ("dptnet_odd", "dprnn_tasnet_odd", "dprnn_tasnet_causal")          # fixtures that must take the composition path


class CountingEmu(EmuBackend):
    """the emulator, remembering which entry points were used"""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "used", set())

    def __getattribute__(self, name):
        attr = object.__getattribute__(self, name)
        if callable(attr) and not name.startswith("_"):
            object.__getattribute__(self, "used").add(name)
        return attr


@pytest.fixture()
def emu():
    backend = CountingEmu()
    old = sepkernels._set_backend_for_tests(backend)
    yield backend
    sepkernels._set_backend_for_tests(old)


def _build(golden_dir, name):
    kind, cfg = SIBLINGS[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    torch.manual_seed(5)
    model = CLASSES[kind](**cfg)
    assert list(model.state_dict().keys()) == list(g["state_keys"]), "state_dict keys / order differ from the reference"
    assert model.num_parameters == int(g["num_parameters"])
    assert sorted(model.get_config().keys()) == sorted(g["config_keys"])
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("positional_encoding.positional_encoding") for k in missing)
    return g, cfg, model.double()


@pytest.mark.parametrize("name", sorted(SIBLINGS))
def test_sibling_separator_matches_the_reference(golden_dir, name, emu):
    g, cfg, model = _build(golden_dir, name)
    on_kernels = name not in COMPOSED
    assert (not model.kernel_path_problems()) == on_kernels
    mixture, sources = torch.from_numpy(g["mixture"]).double(), torch.from_numpy(g["sources"]).double()
    est, latent = model.extract_latent(mixture)
    assert ("decoder_fwd" in emu.used and "encoder_fwd" in emu.used) == on_kernels           # the path under test is the one that ran
    if name == "dptnet":              # features a power of two, gLN everywhere: the token-major stack (models/dptnet.py::_forward_tokens)
        assert "gln_tokens_fwd" in emu.used and "chunk_to_tokens" in emu.used
        assert "attn_fwd" in emu.used                                         # ... and the attention core on sep_attn_* (csrc/attn.hip)
    if name == "galrnet":             # its global attention runs token-major too (models/galr.py::_attend_tokens); the causal one does not
        assert "gln_tokens_fwd" in emu.used and "rownorm_fwd" in emu.used      # (the channel norm in front of it: sep_rownorm_*)
    if name == "sepformer":           # both transformer stacks on token-major rows (models/sepformer.py::_forward_tokens)
        assert "gln_tokens_fwd" in emu.used and "chunk_to_tokens" in emu.used
        assert "rownorm_fwd" in emu.used                                      # ... their layer norms with the residual sums on sep_rownorm_*
    if name == "galrnet_causal":
        assert "gln_tokens_fwd" not in emu.used
    ref = torch.from_numpy(g["output_f64"])
    assert est.shape == ref.shape
    assert (est - ref).abs().max() <= 1e-9 * ref.abs().max()
    assert abs(latent.sum().item() - float(g["latent_f64_sum"])) <= 1e-9 * float(g["latent_f64_abs_sum"])
    loss, pattern = PIT1d(NegSISDR(), n_sources=cfg["n_sources"])(est, sources)
    assert abs(loss.item() - float(g["loss_f64"])) <= 1e-9 * abs(float(g["loss_f64"]))
    assert np.array_equal(pattern.numpy(), g["pattern"])
    loss.backward()
    for k, p in model.named_parameters():
        r = torch.from_numpy(g["grad/" + k]).double()
        assert p.grad is not None, k
        assert (p.grad - r).abs().max() <= 2e-6 * max(r.abs().max().item(), 1e-6), k          # the fixture stores fp64 gradients as fp32


@pytest.mark.parametrize("name", ["dptnet", "galrnet_causal", "sepformer", "sepformer_causal", "dprnn_tasnet_softmax"])
def test_kernel_path_equals_composition(golden_dir, name, emu):
    """same model, same input: the libsepkernels sequence and the module-by-module composition, outputs and gradients"""
    g, cfg, model = _build(golden_dir, name)
    mixture = torch.from_numpy(g["mixture"]).double()
    est_k, latent_k = model._run_kernels(mixture)
    est_c, latent_c = model._run_composed(mixture)
    assert torch.allclose(est_k, est_c.view_as(est_k), rtol=0, atol=1e-12)
    assert torch.allclose(latent_k, latent_c, rtol=0, atol=1e-12)
    (est_k ** 2).sum().backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    (est_c ** 2).sum().backward()
    for k, p in model.named_parameters():
        assert torch.allclose(grads[k], p.grad, rtol=1e-9, atol=1e-12 * max(p.grad.abs().max().item(), 1.0)), k


def test_float32_forward_close_to_the_float64_reference(golden_dir, emu):
    """what the GPU test asserts on the device, with the emulator in fp32"""
    g, cfg, model = _build(golden_dir, "dptnet")
    est = model.float()(torch.from_numpy(g["mixture"]))
    ref = torch.from_numpy(g["output_f64"])
    assert (est.double() - ref).abs().max() <= 2e-4 * ref.abs().max()


def test_multichannel_input_and_error_behaviour(emu):
    kw = dict(kernel_size=4, stride=2, enc_basis="trainable", dec_basis="trainable", enc_nonlinear="relu")
    m = SepFormer(16, in_channels=2, sep_bottleneck_channels=16, sep_chunk_size=8, sep_hop_size=4, sep_num_blocks=1, sep_num_layers_intra=1,
                  sep_num_layers_inter=1, sep_num_heads_intra=2, sep_num_heads_inter=2, sep_d_ff_intra=16, sep_d_ff_inter=16,
                  sep_dropout=0.0, causal=False, n_sources=2, **kw)
    x = torch.randn(1, 1, 2, 101)
    out = m(x)
    assert out.shape == (1, 2, 2, 101)
    assert torch.allclose(out, m._run_composed(x.view(1, 2, 101))[0], atol=1e-5)
    with pytest.raises(ValueError):
        m(torch.randn(1, 101))
    with pytest.raises(ValueError):                               # (B, 1, n_mics, T) is SepFormer's; the reference DPTNet cannot unpack it either
        DPTNet(16, sep_bottleneck_channels=16, sep_hidden_channels=16, sep_chunk_size=8, sep_num_blocks=1, sep_num_heads=2, **kw)(x)
    with pytest.raises(AssertionError):
        DPTNet(18, sep_num_heads=4, **kw)                         # n_basis % sep_num_heads
    with pytest.raises(ValueError):
        GALRNet(16, low_dimension=True, sep_down_chunk_size=None, **kw)
    with pytest.raises(ValueError):
        DPTNet(16, mask_nonlinear="tanh", **kw)
    with pytest.raises(KeyError):
        SepFormer.build_from_pretrained(task="no-such-task")


def test_build_model_round_trip(tmp_path, emu):
    kind, cfg = SIBLINGS["galrnet"]
    model = GALRNet(**cfg)
    package = model.get_config()
    package["state_dict"] = model.state_dict()
    path = os.path.join(str(tmp_path), "last.pth")
    torch.save(package, path)
    again = GALRNet.build_model(path, load_state_dict=True)
    assert again.get_config() == model.get_config()
    x = torch.randn(1, 1, 203)
    assert torch.equal(again.eval()(x), model.eval()(x))


def test_gtu_and_positional_encoding_definitions(emu):
    torch.manual_seed(0)
    gtu = GTU1d(16, 32, kernel_size=1).double()
    x = torch.randn(2, 16, 37, dtype=torch.float64)
    want = torch.tanh(torch.nn.functional.conv1d(x, gtu.map.weight, gtu.map.bias)) * torch.sigmoid(torch.nn.functional.conv1d(x, gtu.map_gate.weight, gtu.map_gate.bias))
    assert torch.allclose(gtu(x), want, atol=1e-12)               # one stacked product on the kernels == the two convolutions
    assert "pw_gemm" in emu.used
    wide = GTU1d(5, 7, kernel_size=3, padding=1)                  # any other geometry: the torch pair
    assert wide(torch.randn(2, 5, 11)).shape == (2, 7, 11)
    pe = PositionalEncoding(8, batch_first=False)
    t, i = 5, 3                                                    # feature 2i: sin(t / 10000^(2i/8)), feature 2i+1: the cosine
    assert abs(pe.positional_encoding[t, 0, 2 * i].item() - np.sin(t / 10000 ** (2 * i / 8))) < 1e-6
    assert abs(pe.positional_encoding[t, 0, 2 * i + 1].item() - np.cos(t / 10000 ** (2 * i / 8))) < 1e-6
    seq = torch.zeros(7, 2, 8)
    assert torch.equal(pe(seq), pe.positional_encoding[:7].expand(7, 2, 8))


def test_layer_norms_split_batches_beyond_one_launch(emu, monkeypatch):
    """the dual-path models normalise thousands of short samples at once; one launch of the gLN / cLN kernels takes 65535 grid
    rows, the modules cut the batch accordingly -- same values, same gradients"""
    import modules.norm as norm
    torch.manual_seed(1)
    gln = norm.GlobalLayerNorm(16).double()
    with torch.no_grad():
        gln.norm.weight.add_(0.2 * torch.randn(16))
        gln.norm.bias.add_(0.2 * torch.randn(16))
    x = torch.randn(11, 16, 5, 7, dtype=torch.float64, requires_grad=True)
    y = gln(x)
    y.square().sum().backward()
    want = (y.detach().clone(), x.grad.clone(), gln.norm.weight.grad.clone(), gln.norm.bias.grad.clone())
    x.grad = None
    gln.zero_grad()
    monkeypatch.setattr(norm, "GRID_ROWS", 16 * 4)                 # four samples per launch: 4 + 4 + 3
    y = gln(x)
    y.square().sum().backward()
    for a, b in zip(want, (y.detach(), x.grad, gln.norm.weight.grad, gln.norm.bias.grad)):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    cln = norm.CumulativeLayerNorm1d(16)
    xf = torch.randn(11, 16, 9)
    monkeypatch.setattr(norm, "GRID_ROWS", 65535)
    whole = cln(xf)
    monkeypatch.setattr(norm, "GRID_ROWS", 4)
    assert torch.equal(whole, cln(xf))


@pytest.mark.parametrize("name", ["dptnet", "sepformer", "galrnet"])
def test_gradient_with_respect_to_the_mixture(golden_dir, emu, name):
    """A mixture that itself requires a gradient (the reference supports it): the kernel path's encoder does not differentiate with respect
    to its input, so such calls take the composition -- and give the same estimate and the input gradient of the composed model."""
    g, _, model = _build(golden_dir, name)
    mixture = torch.from_numpy(g["mixture"]).double().requires_grad_(True)
    est = model(mixture)
    est.square().sum().backward()
    assert mixture.grad is not None and torch.isfinite(mixture.grad).all() and mixture.grad.abs().max() > 0
    with torch.no_grad():
        est_k = model(mixture.detach())              # the kernel path (through the emulator) on the same input
    assert (est_k - est.detach()).abs().max() <= 1e-9 * est.detach().abs().max()


@pytest.mark.parametrize("training,p", [(False, 0.0), (True, 0.0)])
def test_sepformer_feed_forward_on_the_convolution_kernels(emu, training, p):
    """the feed-forward pair of an encoder layer on the 1x1-convolution kernels (widths in multiples of 128: models/sepformer.py::
    _feed_forward_channel_major) against nn.TransformerEncoderLayer's own forward in float64, outputs and every gradient"""
    from models.sepformer import _ChunkPathEncoder, _ff_on_conv_kernels
    torch.manual_seed(3)
    layer = torch.nn.TransformerEncoderLayer(128, 4, 256, dropout=p, activation="relu", batch_first=False).double()
    layer.train(training)
    x = torch.randn(3, 37, 128, dtype=torch.float64, requires_grad=True)             # (N, L, C): 111 tokens, not a multiple of 128
    assert _ff_on_conv_kernels(layer, x)
    y = _ChunkPathEncoder._layer_tokens(layer, x)
    assert "pw_gemm" in emu.used and "pw_wgrad" not in emu.used
    xr = x.detach().clone().requires_grad_(True)
    ref = layer(xr.transpose(0, 1)).transpose(0, 1)                                   # torch's own (T, batch, C) route
    assert (y - ref).abs().max() <= 1e-11 * ref.abs().max()
    w = torch.randn_like(y)
    grads = torch.autograd.grad((y * w).sum(), [x] + list(layer.parameters()))
    refs = torch.autograd.grad((ref * w).sum(), [xr] + list(layer.parameters()))
    assert "pw_wgrad" in emu.used
    for g, r in zip(grads, refs):
        assert (g - r).abs().max() <= 1e-10 * max(r.abs().max().item(), 1e-12)
