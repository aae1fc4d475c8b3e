"""
Minimal wav reader / writer on the standard library (`wave`) + numpy, and an installable stand-in for the three
`torchaudio` calls the reference recipes make (`torchaudio.load(path, frame_offset=, num_frames=)`,
`torchaudio.save(path, tensor, sample_rate=, bits_per_sample=)`, `torchaudio.info`): the image has no torchaudio
 and the wsj0-mix data are plain PCM wav files.

reference call sites: egs/wsj0-mix/common/src/dataset.py:37,81,89 and driver.py:195,202.
"""
import sys
import types
import wave

import numpy as np
import torch

_PCM_SCALE = {1: 128.0, 2: 32768.0, 3: 8388608.0, 4: 2147483648.0}


def wav_info(path):
    """-> (num_frames, num_channels, sample_rate, sample_width_bytes)"""
    with wave.open(path, "rb") as w:
        return w.getnframes(), w.getnchannels(), w.getframerate(), w.getsampwidth()


def read_wav(path, frame_offset=0, num_frames=-1):
    """-> (float32 tensor (channels, frames) in [-1, 1), sample_rate).  PCM 8/16/24/32 bit."""
    with wave.open(path, "rb") as w:
        n, ch, sr, sw = w.getnframes(), w.getnchannels(), w.getframerate(), w.getsampwidth()
        frame_offset = max(0, min(int(frame_offset), n))
        count = n - frame_offset if num_frames is None or num_frames < 0 else min(int(num_frames), n - frame_offset)
        w.setpos(frame_offset)
        raw = w.readframes(count)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32)
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32)
    elif sw == 1:
        x = np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0
    elif sw == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float32)
    else:
        raise ValueError("unsupported sample width {} in {}".format(sw, path))
    x = x.reshape(-1, ch).T / _PCM_SCALE[sw]
    return torch.from_numpy(np.ascontiguousarray(x)), sr


def write_wav(path, signal, sample_rate, bits_per_sample=16):
    """signal: (channels, frames) or (frames,) float tensor in [-1, 1]; written as signed PCM."""
    if bits_per_sample not in (16, 32):
        raise ValueError("bits_per_sample must be 16 or 32")
    x = signal.detach().to("cpu", torch.float32)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    sw = bits_per_sample // 8
    full = _PCM_SCALE[sw]
    q = torch.clamp(torch.round(x.T.contiguous() * full), -full, full - 1).numpy()
    data = q.astype("<i2" if sw == 2 else "<i4").tobytes()
    with wave.open(path, "wb") as w:
        w.setnchannels(x.shape[0])
        w.setsampwidth(sw)
        w.setframerate(int(sample_rate))
        w.writeframes(data)


def install_torchaudio_shim():
    """Registers a `torchaudio` module exposing load / save / info on top of this file (no-op if the real one imports)."""
    try:
        import torchaudio  # noqa: F401
        return False
    except ImportError:
        pass
    shim = types.ModuleType("torchaudio")
    shim.load = lambda path, frame_offset=0, num_frames=-1, **kw: read_wav(path, frame_offset, num_frames)
    shim.save = lambda path, src, sample_rate, bits_per_sample=16, **kw: write_wav(path, src, sample_rate, bits_per_sample)

    class _Info:
        def __init__(self, path):
            self.num_frames, self.num_channels, self.sample_rate, sw = wav_info(path)
            self.bits_per_sample = 8 * sw

    shim.info = _Info
    shim.__version__ = "0.0+sepkernels.shim"
    sys.modules["torchaudio"] = shim
    return True


__all__ = ["read_wav", "write_wav", "wav_info", "install_torchaudio_shim"]
