"""CPU: recipe plumbing around the path: wav I/O + torchaudio stand-in, wsj0-mix style
datasets, trainer (checkpoint format of the reference's driver.py:208-226, resume, overwrite guard, torch-Adam
interchange of the optimizer state) and the variable-length tester.  The kernels are the CPU emulator of the C ABI."""
import argparse
import math
import os
import sys

import pytest
import torch

import sepkernels
from emulator import EmuBackend
from criterion.pit import PIT1d
from criterion.sdr import NegSISDR
from models.conv_tasnet import ConvTasNet
from recipes import audio_io
from recipes.trainer import Tester, Trainer
from recipes.wsj0mix import (EvalDataLoader, TestDataLoader, TrainDataLoader, WaveEvalDataset, WaveTestDataset,
                             WaveTrainDataset, shard_for_rank)
from sepkernels.train import FusedTrainStep

SR = 8000
LENGTHS = {"utt_a": 1000, "utt_b": 700, "utt_c": 450}
TINY = dict(n_basis=16, kernel_size=4, stride=2, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None, causal=False, sep_hidden_channels=32, sep_bottleneck_channels=16, sep_skip_channels=16,
            sep_kernel_size=3, sep_num_blocks=1, sep_num_layers=2, n_sources=2)


@pytest.fixture()
def emu():
    old = sepkernels._set_backend_for_tests(EmuBackend())
    yield
    sepkernels._set_backend_for_tests(old)


@pytest.fixture()
def wav_tree(tmp_path):
    g = torch.Generator().manual_seed(7)
    root = tmp_path / "wav"
    for sub in ("mix", "s1", "s2"):
        (root / sub).mkdir(parents=True)
    for ID, T in LENGTHS.items():
        s = 0.2 * torch.randn(2, T, generator=g)
        audio_io.write_wav(str(root / "s1" / (ID + ".wav")), s[0], SR)
        audio_io.write_wav(str(root / "s2" / (ID + ".wav")), s[1], SR)
        audio_io.write_wav(str(root / "mix" / (ID + ".wav")), s.sum(0), SR)
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(LENGTHS) + "\n")
    return str(root), str(lst)


def test_wav_roundtrip_and_torchaudio_standin(tmp_path):
    x = torch.linspace(-0.9, 0.9, 4001).reshape(1, -1)
    p = str(tmp_path / "a.wav")
    audio_io.write_wav(p, x, SR, bits_per_sample=16)
    y, sr = audio_io.read_wav(p)
    assert sr == SR and y.shape == x.shape and (y - x).abs().max() <= 1.0 / 32768
    y2, _ = audio_io.read_wav(p, frame_offset=100, num_frames=50)
    assert torch.equal(y2, y[:, 100:150])
    assert audio_io.wav_info(p)[:3] == (4001, 1, SR)
    stereo = torch.stack([x[0], -x[0]])
    audio_io.write_wav(p, stereo, 44100, bits_per_sample=32)
    y3, sr3 = audio_io.read_wav(p)
    assert sr3 == 44100 and y3.shape == (2, 4001) and (y3 - stereo).abs().max() < 1e-6
    had = "torchaudio" in sys.modules
    installed = audio_io.install_torchaudio_shim()
    try:
        import torchaudio
        w, sr4 = torchaudio.load(p, frame_offset=10, num_frames=5)
        assert w.shape == (2, 5) and sr4 == 44100
        torchaudio.save(p, x, sample_rate=SR, bits_per_sample=16)
        assert torchaudio.info(p).num_frames == 4001
    finally:
        if installed and not had:
            del sys.modules["torchaudio"]


def test_datasets_follow_the_reference_segmentation(wav_tree):
    root, lst = wav_tree
    samples, overlap = 256, 128
    tr = WaveTrainDataset(root, lst, samples=samples, overlap=overlap, n_sources=2)
    expect = sum((T - samples) // (samples - overlap) + 1 for T in LENGTHS.values() if T >= samples)
    assert len(tr) == expect
    m, s = tr[3]
    assert m.shape == (1, samples) and s.shape == (2, samples)
    assert (m[0] - s.sum(0)).abs().max() <= 2.5 / 32768          # mixture = s1 + s2 up to 16-bit rounding
    ev = WaveEvalDataset(root, lst, max_samples=800, n_sources=2)
    assert len(ev) == 3
    m, s, ID = ev[0]
    assert ID == "utt_a" and m.shape == (1, 800) and s.shape == (2, 800)
    assert ev[2][0].shape == (1, 450)
    with pytest.raises(AssertionError):
        EvalDataLoader(ev, batch_size=2)
    b = next(iter(TestDataLoader(WaveTestDataset(root, lst, n_sources=2), batch_size=1)))
    assert b[0].shape == (1, 1, 1000) and b[1].shape == (1, 2, 1000) and b[2] == ["utt_a"]
    shards = [shard_for_rank(tr, r, 2, seed=3) for r in range(2)]
    assert len(shards[0]) == len(shards[1]) == len(tr) // 2
    assert not set(shards[0].indices) & set(shards[1].indices)


def _args(tmp_path, **kw):
    d = dict(model_dir=str(tmp_path / "model"), loss_dir=str(tmp_path / "loss"), sample_dir=str(tmp_path / "sample"), epochs=2,
             lr=1e-3, max_norm=5.0, continue_from=None, overwrite=False, sample_rate=SR, weight_decay=0.0)
    d.update(kw)
    return argparse.Namespace(**d)


def _loaders(wav_tree):
    root, lst = wav_tree
    tr = WaveTrainDataset(root, lst, samples=256, overlap=128, n_sources=2)
    ev = WaveEvalDataset(root, lst, max_samples=600, n_sources=2)
    return {"train": TrainDataLoader(tr, batch_size=3, shuffle=False, drop_last=True), "valid": EvalDataLoader(ev, batch_size=1)}


def test_trainer_checkpoint_format_resume_and_overwrite_guard(tmp_path, wav_tree, emu):
    torch.manual_seed(0)
    model = ConvTasNet(**TINY)
    crit = PIT1d(NegSISDR(), n_sources=2)
    tr = Trainer(model, _loaders(wav_tree), crit, _args(tmp_path))
    tr.run()
    assert all(math.isfinite(v) for v in tr.train_loss.tolist() + tr.valid_loss.tolist())
    for name in ("best.pth", "last.pth"):
        assert os.path.exists(os.path.join(tmp_path, "model", name))
    assert os.path.exists(os.path.join(tmp_path, "sample", "utt_a", "mixture.wav"))
    ck = torch.load(os.path.join(tmp_path, "model", "last.pth"), weights_only=False)
    # reference driver.py:208-226: model config + these seven entries
    assert set(ck) == set(model.get_config()) | {"state_dict", "optim_dict", "best_loss", "no_improvement", "train_loss", "valid_loss", "epoch"}
    assert ck["epoch"] == 2 and set(ck["state_dict"]) == set(model.state_dict())
    # the optimizer state is a valid torch.optim.Adam state_dict and carries the fused moments
    twin = ConvTasNet(**TINY)
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    opt.load_state_dict(ck["optim_dict"])
    spans = tr.step._spans()
    for i, (off, n, shape) in enumerate(spans):
        assert torch.equal(opt.state_dict()["state"][i]["exp_avg"].reshape(-1), tr.step.m[off:off + n])
    # refuse to clobber, then resume for one more epoch
    with pytest.raises(ValueError):
        Trainer(ConvTasNet(**TINY), _loaders(wav_tree), crit, _args(tmp_path))
    model2 = ConvTasNet(**TINY)
    tr2 = Trainer(model2, _loaders(wav_tree), crit, _args(tmp_path, epochs=3, continue_from=os.path.join(tmp_path, "model", "last.pth")))
    assert tr2.start_epoch == 2 and tr2.step.step_count == tr.step.step_count
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(tr2.step.m, tr.step.m) and torch.equal(tr2.step.v, tr.step.v)
    tr2.run()
    assert torch.load(os.path.join(tmp_path, "model", "last.pth"), weights_only=False)["epoch"] == 3


def test_torch_adam_state_resumes_the_fused_step(emu):
    """A checkpoint written by the reference (torch.optim.Adam state) continues bit-for-bit in the fused step."""
    torch.manual_seed(1)
    a, b = ConvTasNet(**TINY), ConvTasNet(**TINY)
    b.load_state_dict(a.state_dict())
    crit = PIT1d(NegSISDR(), n_sources=2)
    mix = 0.3 * torch.randn(2, 1, 200)
    src = 0.3 * torch.randn(2, 2, 200)
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    for _ in range(2):                                   # two reference-style steps (driver.py:146-155) on model a
        loss, _ = crit(a(mix), src)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(a.parameters(), 5.0)
        opt.step()
    fused = FusedTrainStep(b, crit, lr=1e-3, max_norm=5.0)
    fused(mix, src)
    fused(mix, src)
    for (k, pa), (_, pb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert (pa - pb).abs().max() <= 1e-6 * (1 + pa.abs().max()), k
    # hand the torch optimizer's state to a fresh fused step and take a third step on both sides
    c = ConvTasNet(**TINY)
    c.load_state_dict(a.state_dict())
    fused_c = FusedTrainStep(c, crit, lr=1e-3, max_norm=5.0)
    fused_c.load_optim_state_dict(opt.state_dict())
    assert fused_c.step_count == 2
    loss, _ = crit(a(mix), src)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(a.parameters(), 5.0)
    opt.step()
    fused_c(mix, src)
    for (k, pa), (_, pc) in zip(a.state_dict().items(), c.state_dict().items()):
        assert (pa - pc).abs().max() <= 1e-6 * (1 + pa.abs().max()), k


def test_adam_state_of_a_fourier_model_round_trips_with_torch_adam(emu):
    """round-4 advisor finding: a Fourier filterbank carries integer `time_seq` parameters that torch.optim.Adam(model.parameters()) counts
    in its state indices; the fused step's optim_dict has to use the same numbering in both directions."""
    cfg = dict(TINY, enc_basis="trainableFourierTrainablePhase", dec_basis="trainableFourierTrainablePhase", enc_nonlinear=None, window_fn="hann",
               enc_onesided=False, enc_return_complex=False)
    torch.manual_seed(4)
    a = ConvTasNet(**cfg)
    n_all = len(list(a.parameters()))
    n_float = len([p for p in a.parameters() if p.is_floating_point()])
    assert n_all > n_float                                           # the integer parameters exist in this configuration
    crit = PIT1d(NegSISDR(), n_sources=2)
    mix, src = 0.3 * torch.randn(2, 1, 200), 0.3 * torch.randn(2, 2, 200)
    fused = FusedTrainStep(a, crit, lr=1e-3, max_norm=5.0)
    fused(mix, src)
    sd = fused.optim_state_dict()
    assert sd["param_groups"][0]["params"] == list(range(n_all)) and len(sd["state"]) == n_float
    # torch's optimizer over ALL parameters accepts it ...
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    opt.load_state_dict(sd)
    params = list(a.parameters())
    for i, st in opt.state_dict()["state"].items():
        assert params[i].is_floating_point() and st["exp_avg"].shape == params[i].shape
    # ... and what torch writes comes back to the same moments
    b = ConvTasNet(**cfg)
    b.load_state_dict(a.state_dict())
    fused_b = FusedTrainStep(b, crit, lr=1e-3, max_norm=5.0)
    fused_b.load_optim_state_dict(opt.state_dict())
    assert fused_b.step_count == 1 and torch.equal(fused_b.m, fused.m) and torch.equal(fused_b.v, fused.v)


def test_tester_variable_length_inference(tmp_path, wav_tree, emu):
    root, lst = wav_tree
    torch.manual_seed(2)
    model = ConvTasNet(**TINY)
    loader = TestDataLoader(WaveTestDataset(root, lst, n_sources=2), batch_size=1)
    args = argparse.Namespace(sample_rate=SR, n_sources=2, out_dir=str(tmp_path / "out"), model_path=None)
    res = Tester(model, loader, PIT1d(NegSISDR(), n_sources=2), args).run()
    assert all(math.isfinite(v) for v in res.values())
    assert os.path.exists(os.path.join(tmp_path, "out", "utt_c_2-estimated.wav"))


def test_training_trajectory_follows_the_reference_trainer_through_the_emulator(golden_dir, emu):
    """CPU form of tests/test_gpu_recipe.py::test_training_trajectory_follows_the_reference_trainer: the same eight steps of the recipe's
    step through the C-ABI emulator in fp64 against the reference's fp64 trajectory (1e-7: Adam amplifies nothing in eight steps)."""
    import numpy as np
    from oracle.make_golden import TRAJ, TRAJ_CFG, traj_batches
    from sepkernels.train import FusedTrainStep
    fx = np.load(os.path.join(golden_dir, "train_trajectory.npz"))
    torch.manual_seed(TRAJ["model_seed"])
    model = ConvTasNet(**TRAJ_CFG).double()
    step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=TRAJ["lr"], max_norm=TRAJ["max_norm"])
    losses = [step(mix.double(), src.double()).item() for mix, src in traj_batches()]
    assert np.abs(np.array(losses) - fx["loss_f64"]).max() <= 1e-7 * np.abs(fx["loss_f64"]).max(), losses
    for k, v in model.state_dict().items():
        want = fx["pfp/" + k]
        assert abs(v.double().sum().item() - want[0]) <= 1e-6 * want[1] + 1e-12, k
