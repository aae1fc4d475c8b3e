"""
wsj0-mix style datasets and loaders for time-domain separation.

Directory contract (same as the reference, egs/wsj0-mix/common/src/dataset.py:13-153): `wav_root/mix/<ID>.wav`,
`wav_root/s1/<ID>.wav` ... `wav_root/s<n>/<ID>.wav`, and a list file with one `<ID>` per line.

* WaveTrainDataset  fixed-length training segments: windows of `samples` frames every `samples - overlap` frames
                    (overlap defaults to samples // 2), incomplete tail windows dropped  -> (mixture (1,T), sources (n,T))
* WaveEvalDataset   one item per utterance, cut to `max_samples`                        -> (+ ID)
* WaveTestDataset   same items, for the tester                                          -> (+ ID)
* TrainDataLoader / EvalDataLoader / TestDataLoader   thin DataLoader subclasses (eval / test insist on batch_size 1,
                    utterances have different lengths)
* DevicePrefetcher  wraps a loader: pinned staging + asynchronous H2D on a copy stream, one batch ahead, so the fused
                    step is not starved by wav decoding (the reference loads synchronously with num_workers=0 and
                    `.cuda()` in the loop, driver.py:141-144)
* shard_for_rank    equal-size index shards for one-process-per-GPU data parallelism
"""
import os

import torch
from torch.utils.data import DataLoader, Dataset, Subset

from .audio_io import read_wav, wav_info


def _read_ids(list_path):
    with open(list_path) as f:
        return [line.strip() for line in f if line.strip()]


class _WaveItems(Dataset):
    """items: list of (ID, start, end); every item reads the same frame range of mix and s1..sn."""

    def __init__(self, wav_root, list_path, n_sources):
        super().__init__()
        self.wav_root = os.path.abspath(wav_root)
        self.list_path = os.path.abspath(list_path)
        self.n_sources = n_sources
        self.items = []

    def _path(self, sub, ID):
        return os.path.join(self.wav_root, sub, ID + ".wav")

    def _load(self, idx):
        ID, start, end = self.items[idx]
        mixture, _ = read_wav(self._path("mix", ID), start, end - start)
        sources = torch.cat([read_wav(self._path("s{}".format(k + 1), ID), start, end - start)[0] for k in range(self.n_sources)], dim=0)
        return mixture, sources, ID, start, end

    def __len__(self):
        return len(self.items)


class WaveTrainDataset(_WaveItems):
    def __init__(self, wav_root, list_path, samples=32000, overlap=None, n_sources=2):
        super().__init__(wav_root, list_path, n_sources)
        hop = samples - (samples // 2 if overlap is None else overlap)
        if hop <= 0:
            raise ValueError("overlap must be smaller than samples")
        for ID in _read_ids(self.list_path):
            total = wav_info(self._path("mix", ID))[0]
            for start in range(0, total - samples + 1, hop):
                self.items.append((ID, start, start + samples))

    def __getitem__(self, idx):
        mixture, sources, _, _, _ = self._load(idx)
        return mixture, sources


class WaveEvalDataset(_WaveItems):
    def __init__(self, wav_root, list_path, max_samples=None, n_sources=2):
        super().__init__(wav_root, list_path, n_sources)
        for ID in _read_ids(self.list_path):
            total = wav_info(self._path("mix", ID))[0]
            self.items.append((ID, 0, total if max_samples is None else min(total, max_samples)))

    def __getitem__(self, idx):
        mixture, sources, ID, _, _ = self._load(idx)
        return mixture, sources, ID


class WaveTestDataset(WaveEvalDataset):
    pass


class TrainDataLoader(DataLoader):
    pass


def _utterance_collate(batch):
    mixture = torch.stack([b[0] for b in batch], dim=0)
    sources = torch.stack([b[1] for b in batch], dim=0)
    return mixture, sources, [b[2] for b in batch]


class EvalDataLoader(DataLoader):
    def __init__(self, *args, **kwargs):
        kwargs.setdefault("collate_fn", _utterance_collate)
        super().__init__(*args, **kwargs)
        if self.batch_size != 1:
            raise AssertionError("batch_size is expected 1, but given {}".format(self.batch_size))


class TestDataLoader(EvalDataLoader):
    __test__ = False        # (not a pytest class)


def shard_for_rank(dataset, rank, world, seed=0, shuffle=True, epoch=0):
    """Equal shards (the tail that does not divide is dropped so every rank runs the same number of steps)."""
    n = len(dataset)
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    per = n // world
    return Subset(dataset, order[rank * per:(rank + 1) * per])


class DevicePrefetcher:
    """for batch in DevicePrefetcher(loader, device): tensors arrive on `device`, next batch's H2D already in flight."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        if not self.on_gpu:
            return batch
        out = []
        with torch.cuda.stream(self.copy_stream):
            for x in batch:
                out.append(x.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(x) else x)
        return out

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            if self.on_gpu:
                torch.cuda.current_stream(self.device).wait_stream(self.copy_stream)
                for x in nxt:
                    if torch.is_tensor(x):
                        x.record_stream(torch.cuda.current_stream(self.device))
            cur = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            yield tuple(cur)
