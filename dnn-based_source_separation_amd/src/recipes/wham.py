"""WHAM! (noisy wsj0-2mix) wave datasets for the Conv-TasNet recipes.  Directory contract and item
semantics of reference egs/wham/common/src/dataset.py:20-218: `<wav_root>/{mix_single,mix_both,s1,s2,noise}/<ID>.wav`, one ID
per line of the list file; task 'enhance' (mix_single for one speaker, mix_both for two) or 'separate-noisy' (mix_both, two
speakers).  Training items are fixed-length windows (`samples`, hop `samples - overlap`, a window that would pass the end of the
mixture is dropped), evaluation items are whole utterances cut at `max_samples`.  Reads go through recipes.audio_io (no
torchaudio in the image); loaders and the rank sharding are those of recipes.wsj0mix."""
import os

import torch
from torch.utils.data import Dataset

from .audio_io import read_wav, wav_info
from .wsj0mix import EvalDataLoader, TestDataLoader, TrainDataLoader, _read_ids     # noqa: F401  (re-exported)


def _mix_type(task, n_sources):
    if task == "enhance":
        if n_sources == 1:
            return "single"
        if n_sources == 2:
            return "both"
        raise ValueError("n_sources is expected 1 or 2 in enhancement task, but given {}.".format(n_sources))
    if task == "separate-noisy":
        if n_sources == 2:
            return "both"
        raise ValueError("n_sources is expected 2 in separation task, but given {}.".format(n_sources))
    raise ValueError("`task` is expected 'enhance' or 'separate-noisy', but given {}.".format(task))


class WaveDataset(Dataset):
    """items (ID, start, end): the same frame range of the mixture, s1..sn and the noise"""

    def __init__(self, wav_root, list_path, task="separate-noisy", n_sources=2):
        super().__init__()
        self.wav_root, self.list_path = os.path.abspath(wav_root), os.path.abspath(list_path)
        self.task, self.n_sources = task, n_sources
        self.mix_type = _mix_type(task, n_sources)
        self.items = []

    def _path(self, sub, ID):
        return os.path.join(self.wav_root, sub, ID + ".wav")

    def _total(self, ID):
        return wav_info(self._path("mix_" + self.mix_type, ID))[0]

    def load(self, idx):
        """-> mixture (1, T), sources (n_sources, T), noise (1, T), segment ID '<ID>_<start>-<end>'"""
        ID, start, end = self.items[idx]
        n = end - start
        sources = torch.cat([read_wav(self._path("s{}".format(k + 1), ID), start, n)[0] for k in range(self.n_sources)], dim=0)
        noise, _ = read_wav(self._path("noise", ID), start, n)
        mixture, _ = read_wav(self._path("mix_" + self.mix_type, ID), start, n)
        return mixture, sources, noise, "{}_{}-{}".format(ID, start, end)

    def __len__(self):
        return len(self.items)


class WaveTrainDataset(WaveDataset):
    def __init__(self, wav_root, list_path, task="separate-noisy", samples=32000, overlap=None, n_sources=2):
        super().__init__(wav_root, list_path, task=task, n_sources=n_sources)
        hop = samples - (samples // 2 if overlap is None else overlap)
        if hop <= 0:
            raise ValueError("overlap must be smaller than samples")
        for ID in _read_ids(self.list_path):
            total = self._total(ID)
            for start in range(0, total, hop):
                if start + samples > total:
                    break
                self.items.append((ID, start, start + samples))

    def __getitem__(self, idx):
        mixture, sources, _, _ = self.load(idx)
        return mixture, sources


class WaveEvalDataset(WaveDataset):
    def __init__(self, wav_root, list_path, task="separate-noisy", max_samples=None, n_sources=2):
        super().__init__(wav_root, list_path, task=task, n_sources=n_sources)
        for ID in _read_ids(self.list_path):
            total = self._total(ID)
            self.items.append((ID, 0, total if max_samples is None else min(total, max_samples)))

    def __getitem__(self, idx):
        mixture, sources, _, _ = self.load(idx)
        return mixture, sources, self.items[idx][0]


class WaveTestDataset(WaveEvalDataset):
    __test__ = False        # (not a pytest class)
