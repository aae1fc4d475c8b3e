"""MUSDB18 / MUSDB18-HQ wave datasets for the music Conv-TasNet recipe (in_channels = 2, 44.1 kHz,
sources bass / drums / other / vocals in a FIXED order, so the criterion is a plain distance, no PIT).  Directory contract and
item semantics of reference egs/musdb18/common/src/dataset.py:15-271: `<root>/train/<track>/{mixture,bass,drums,other,vocals}.wav`
(decoded stems), `<root>/train.txt` and `<root>/validation.txt` with one track name per line.  The mixture is mixture.wav when
all four sources take part and the sum of the chosen stems otherwise; with a list `target` an item is
(mixture (1, n_mics, T), target (len(target), n_mics, T)), with a single name (n_mics, T) each.  Reads go through
recipes.audio_io; the loaders are those of recipes.wsj0mix."""
import os

import torch
from torch.utils.data import Dataset

from .audio_io import read_wav, wav_info
from .wsj0mix import TrainDataLoader, _read_ids     # noqa: F401  (re-exported)

SOURCES = ["bass", "drums", "other", "vocals"]
SAMPLE_RATE_MUSDB18 = 44100


class WaveDataset(Dataset):
    def __init__(self, musdb18_root, sample_rate=SAMPLE_RATE_MUSDB18, sources=SOURCES, target=None):
        super().__init__()
        if sample_rate != SAMPLE_RATE_MUSDB18:
            raise AssertionError("sample rate is expected {}, but given {}".format(SAMPLE_RATE_MUSDB18, sample_rate))
        sources = list(sources)
        if target is None:
            target = sources
        for t in (target if isinstance(target, list) else [target]):
            if t not in sources:
                raise AssertionError("`sources` doesn't contain target {}".format(t))
        self.musdb18_root = os.path.abspath(musdb18_root)
        self.sources, self.target = sources, target
        self.tracks, self.items = [], []            # tracks: names; items: (track index, start, samples)

    def _path(self, name, stem):
        return os.path.join(self.musdb18_root, "train", name, stem + ".wav")

    def _add_track(self, name):
        frames, _, sr, _ = wav_info(self._path(name, "mixture"))
        if sr != SAMPLE_RATE_MUSDB18:
            raise AssertionError("{}: sample rate {}".format(name, sr))
        self.tracks.append(name)
        return len(self.tracks) - 1, frames

    def load(self, idx):
        track, start, samples = self.items[idx]
        name = self.tracks[track]
        if set(self.sources) == set(SOURCES):
            mixture, _ = read_wav(self._path(name, "mixture"), start, samples)
        else:
            mixture = torch.stack([read_wav(self._path(name, s), start, samples)[0] for s in self.sources], dim=0).sum(dim=0)
        if isinstance(self.target, list):
            target = torch.stack([read_wav(self._path(name, s), start, samples)[0] for s in self.target], dim=0)
            mixture = mixture.unsqueeze(0)
        else:
            target, _ = read_wav(self._path(name, self.target), start, samples)
        return mixture, target, name

    def __len__(self):
        return len(self.items)


class WaveTrainDataset(WaveDataset):
    def __init__(self, musdb18_root, sample_rate=SAMPLE_RATE_MUSDB18, samples=4 * SAMPLE_RATE_MUSDB18, overlap=None, sources=SOURCES, target=None, include_valid=False):
        super().__init__(musdb18_root, sample_rate=sample_rate, sources=sources, target=target)
        valid = set(_read_ids(os.path.join(self.musdb18_root, "validation.txt")))
        hop = samples - (samples // 2 if overlap is None else overlap)
        if hop <= 0:
            raise ValueError("overlap must be smaller than samples")
        for name in _read_ids(os.path.join(self.musdb18_root, "train.txt")):
            if not include_valid and name in valid:
                continue
            track, frames = self._add_track(name)
            for start in range(0, frames, hop):
                if start + samples >= frames:
                    break
                self.items.append((track, start, samples))

    def __getitem__(self, idx):
        mixture, target, _ = self.load(idx)
        return mixture, target


class WaveEvalDataset(WaveDataset):
    def __init__(self, musdb18_root, sample_rate=SAMPLE_RATE_MUSDB18, max_samples=4 * SAMPLE_RATE_MUSDB18, sources=SOURCES, target=None):
        super().__init__(musdb18_root, sample_rate=sample_rate, sources=sources, target=target)
        self.max_samples = max_samples
        for name in _read_ids(os.path.join(self.musdb18_root, "validation.txt")):
            track, frames = self._add_track(name)
            self.items.append((track, 0, frames if max_samples is None else min(frames, max_samples)))

    def __getitem__(self, idx):
        mixture, target, _ = self.load(idx)
        return mixture, target
