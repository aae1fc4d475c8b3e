"""CPU: the WHAM and MUSDB18 wave datasets (recipes/wham.py, recipes/musdb18.py) against the LIVE reference dataset classes
(egs/wham/common/src/dataset.py, egs/musdb18/common/src/dataset.py) on the same synthetic wav trees: same number of items, same
tensors, same IDs.  The reference reads through `torchaudio`; the image has none, so it gets recipes.audio_io's stand-in."""
import importlib.util
import os
import sys
import types

import pytest
import torch

from recipes import musdb18 as M
from recipes import wham as W
from recipes.audio_io import read_wav, wav_info, write_wav

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _load_reference_module(path, name, extra=()):
    """import one reference file under a private name, with a torchaudio stand-in and stub modules for what it imports but does
    not use on the wave path"""
    shim = types.ModuleType("torchaudio")
    shim.load = lambda p, frame_offset=0, num_frames=-1, **kw: read_wav(p, frame_offset, num_frames)

    class _Info:
        def __init__(self, p):
            self.num_frames, self.num_channels, self.sample_rate, _ = wav_info(p)
    shim.info = _Info
    saved = {k: sys.modules.get(k) for k in ("torchaudio",) + tuple(extra)}
    sys.modules["torchaudio"] = shim
    for k in extra:
        m = types.ModuleType(k)
        m.build_window = m.stft = None
        sys.modules[k] = m
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _wav(path, channels, frames, seed, sr):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    write_wav(path, 0.3 * torch.randn(channels, frames, generator=g), sr)


def test_wham_datasets_equal_the_reference(tmp_path):
    ref = _load_reference_module(os.path.join(REF, "egs/wham/common/src/dataset.py"), "ref_wham_dataset")
    root, lens = str(tmp_path / "wham"), {"a01": 9000, "b02": 4000, "c03": 6500}
    for i, (ID, n) in enumerate(lens.items()):
        for j, sub in enumerate(["mix_single", "mix_both", "s1", "s2", "noise"]):
            _wav(os.path.join(root, sub, ID + ".wav"), 1, n, 10 * i + j, 8000)
    lst = str(tmp_path / "ids.txt")
    open(lst, "w").write("\n".join(lens) + "\n")
    for task, n_src in (("separate-noisy", 2), ("enhance", 1), ("enhance", 2)):
        r = ref.WaveTrainDataset(root, lst, task=task, samples=4000, overlap=1000, n_sources=n_src)
        m = W.WaveTrainDataset(root, lst, task=task, samples=4000, overlap=1000, n_sources=n_src)
        assert len(m) == len(r) > 0
        for i in range(len(r)):
            (mr, sr_), (mm, sm) = r[i], m[i]
            assert torch.equal(mr, mm) and torch.equal(sr_, sm)
        r = ref.WaveEvalDataset(root, lst, task=task, max_samples=5000, n_sources=n_src)
        m = W.WaveEvalDataset(root, lst, task=task, max_samples=5000, n_sources=n_src)
        assert len(m) == len(r) == 3
        for i in range(3):
            assert torch.equal(r[i][0], m[i][0]) and torch.equal(r[i][1], m[i][1]) and r[i][2] == m[i][2]
    # the exact window rule: a window ending exactly at the end of the file is kept (4000-sample file, samples=4000)
    assert any(it[0] == "b02" for it in W.WaveTrainDataset(root, lst, samples=4000, overlap=1000).items)
    with pytest.raises(ValueError):
        W.WaveTrainDataset(root, lst, task="separate-noisy", n_sources=3)
    with pytest.raises(ValueError):
        W.WaveTrainDataset(root, lst, task="denoise")
    loader = W.TrainDataLoader(W.WaveTrainDataset(root, lst, samples=4000, overlap=1000), batch_size=2, shuffle=False)
    mixture, sources = next(iter(loader))
    assert mixture.shape == (2, 1, 4000) and sources.shape == (2, 2, 4000)


def test_musdb18_datasets_equal_the_reference(tmp_path):
    ref = _load_reference_module(os.path.join(REF, "egs/musdb18/common/src/dataset.py"), "ref_musdb18_dataset",
                                 extra=("utils", "utils.audio", "transforms", "transforms.stft"))
    root, sr = str(tmp_path / "musdb18"), 44100
    tracks = {"Artist A - One": 30000, "Artist B - Two": 21000, "Artist C - Three": 26000}
    for i, (name, n) in enumerate(tracks.items()):
        for j, stem in enumerate(["mixture"] + M.SOURCES):
            _wav(os.path.join(root, "train", name, stem + ".wav"), 2, n, 100 * i + j, sr)
    open(os.path.join(root, "train.txt"), "w").write("\n".join(tracks) + "\n")
    open(os.path.join(root, "validation.txt"), "w").write("Artist B - Two\n")
    for sources, target in ((M.SOURCES, None), (M.SOURCES, "vocals"), (["drums", "vocals"], ["vocals"]), (["bass", "drums", "other"], None)):
        r = ref.WaveTrainDataset(root, sample_rate=sr, samples=8000, overlap=2000, sources=sources, target=target)
        m = M.WaveTrainDataset(root, sample_rate=sr, samples=8000, overlap=2000, sources=sources, target=target)
        assert len(m) == len(r) > 0
        for i in range(len(r)):
            (xr, tr), (xm, tm) = r[i], m[i]
            assert xr.shape == xm.shape and tr.shape == tm.shape
            assert torch.equal(xr, xm) and torch.equal(tr, tm)
        r = ref.WaveEvalDataset(root, sample_rate=sr, max_samples=10000, sources=sources, target=target)
        m = M.WaveEvalDataset(root, sample_rate=sr, max_samples=10000, sources=sources, target=target)
        assert len(m) == len(r) == 1 and torch.equal(r[0][0], m[0][0]) and torch.equal(r[0][1], m[0][1])
    full = M.WaveTrainDataset(root, samples=8000, overlap=2000, include_valid=True)
    assert len(full) > len(M.WaveTrainDataset(root, samples=8000, overlap=2000))
    x, t = full[0]
    assert x.shape == (1, 2, 8000) and t.shape == (4, 2, 8000)            # what ConvTasNet(in_channels=2, n_sources=4) trains on
    with pytest.raises(AssertionError):
        M.WaveTrainDataset(root, sources=["drums"], target="vocals")
