"""GPU: the recipe plumbing on the real kernels -- a short training run from wav files (pinned async prefetch, fused
step, validation at B = 1 with variable lengths, checkpoint) and the tester."""
import argparse
import math
import os

import pytest
import torch

from criterion.pit import PIT1d
from criterion.sdr import NegSISDR
from models.conv_tasnet import ConvTasNet
from recipes import audio_io
from recipes.trainer import Tester, Trainer
from recipes.wsj0mix import EvalDataLoader, TestDataLoader, TrainDataLoader, WaveEvalDataset, WaveTestDataset, WaveTrainDataset

pytestmark = pytest.mark.gpu
SR = 8000
CFG = dict(n_basis=64, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None, causal=False,
           sep_hidden_channels=128, sep_bottleneck_channels=64, sep_skip_channels=64, sep_kernel_size=3, sep_num_blocks=1,
           sep_num_layers=3, n_sources=2)


def _tree(tmp_path):
    g = torch.Generator().manual_seed(5)
    root = tmp_path / "wav"
    for sub in ("mix", "s1", "s2"):
        (root / sub).mkdir(parents=True)
    ids = {"a": 9000, "b": 12345, "c": 8000, "d": 10001}
    for ID, T in ids.items():
        s = 0.2 * torch.randn(2, T, generator=g)
        for k in range(2):
            audio_io.write_wav(str(root / "s{}".format(k + 1) / (ID + ".wav")), s[k], SR)
        audio_io.write_wav(str(root / "mix" / (ID + ".wav")), s.sum(0), SR)
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(ids) + "\n")
    return str(root), str(lst)


def test_train_validate_checkpoint_test_on_gpu(tmp_path):
    root, lst = _tree(tmp_path)
    torch.manual_seed(3)
    model = ConvTasNet(**CFG).cuda()
    crit = PIT1d(NegSISDR(), n_sources=2)
    loaders = {"train": TrainDataLoader(WaveTrainDataset(root, lst, samples=4000, overlap=2000), batch_size=4, shuffle=True, drop_last=True),
               "valid": EvalDataLoader(WaveEvalDataset(root, lst, max_samples=10000), batch_size=1)}
    args = argparse.Namespace(model_dir=str(tmp_path / "model"), loss_dir=str(tmp_path / "loss"), sample_dir=str(tmp_path / "sample"),
                              epochs=3, lr=1e-3, max_norm=5.0, continue_from=None, overwrite=False, sample_rate=SR, weight_decay=0.0)
    tr = Trainer(model, loaders, crit, args)
    tr.run()
    losses = tr.train_loss.tolist()
    assert all(math.isfinite(v) for v in losses + tr.valid_loss.tolist())
    assert losses[-1] < losses[0]                              # three epochs on random sources still reduce the training loss
    ck = torch.load(os.path.join(tmp_path, "model", "last.pth"), weights_only=False)
    assert ck["epoch"] == 3 and all(v.device.type == "cpu" for v in ck["state_dict"].values())
    targs = argparse.Namespace(sample_rate=SR, n_sources=2, out_dir=str(tmp_path / "out"), model_path=os.path.join(tmp_path, "model", "best.pth"))
    res = Tester(ConvTasNet(**CFG).cuda(), TestDataLoader(WaveTestDataset(root, lst), batch_size=1), crit, targs).run()
    assert all(math.isfinite(v) for v in res.values())


def test_training_trajectory_follows_the_reference_trainer(golden_dir):
    """The recipe's own step (sepkernels.train.FusedTrainStep: what recipes.trainer.Trainer drives -- fused forward / PIT / backward, clip and
    Adam on flat buffers) against the REFERENCE's training step (driver.py:141-157 with torch.optim.Adam and clip_grad_norm_, run by
    oracle/make_golden.py::train_trajectory_golden on the unmodified reference classes in fp64): the same seeded model, the same eight
    batches -- every step's loss within 1e-3, the parameters afterwards within 1e-3 of their scale."""
    import numpy as np
    from oracle.make_golden import TRAJ, TRAJ_CFG, traj_batches
    from sepkernels.train import FusedTrainStep
    fx = np.load(os.path.join(golden_dir, "train_trajectory.npz"))
    torch.manual_seed(TRAJ["model_seed"])
    model = ConvTasNet(**TRAJ_CFG).cuda()
    assert model.fused
    step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=TRAJ["lr"], max_norm=TRAJ["max_norm"])
    losses = [step(mix.cuda(), src.cuda()).item() for mix, src in traj_batches()]
    ref = fx["loss_f64"]
    rel = np.abs(np.array(losses) - ref) / np.abs(ref)
    assert rel.max() <= 1e-3, (losses, ref.tolist())
    assert rel[0] <= 1e-5                                               # the first step is a pure forward comparison
    for k, v in model.state_dict().items():
        want = fx["pfp/" + k]
        got = np.array([v.double().sum().item(), v.double().abs().sum().item(), v.double().abs().max().item()])
        assert abs(got[1] - want[1]) <= 1e-3 * want[1] + 1e-9 and abs(got[2] - want[2]) <= 1e-3 * want[2] + 1e-9, k
