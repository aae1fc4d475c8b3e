"""Which Python lines of the headline step issue device-to-device copies / fills (torch.profiler with stacks, one eager step):
    python tools/find_copies.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                                          # noqa: E402
import sepkernels                                                          # noqa: E402
from sepkernels.train import FusedTrainStep                                # noqa: E402
from models.conv_tasnet import ConvTasNet                                  # noqa: E402
from criterion.sdr import NegSISDR                                         # noqa: E402
from criterion.pit import PIT1d                                            # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(111)
model = ConvTasNet(**dict(B.PAPER)).to(dev)
step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=1e-3, max_norm=5.0)
src = (0.1 * torch.randn(16, 2, B.T_SAMPLES)).to(dev)
mix = src.sum(1, keepdim=True).contiguous()
for _ in range(3):
    step(mix, src)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

count = collections.Counter()
WATCH = ("copy_", "fill_", "zero_", "clone", "_to_copy", "zeros", "full", "ones", "cat", "stack", "add", "mul", "sum", "neg", "mean", "abs", "amax", "index", "gather")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            st = [f for f in traceback.extract_stack() if "dnn-based_source_separation_amd" in f.filename or f.filename.endswith("bench.py")]
            where = "{}:{} {}".format(os.path.basename(st[-1].filename), st[-1].lineno, st[-1].line) if st else "?"
            count[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step(mix, src)
torch.cuda.synchronize()
for (name, where), n in count.most_common(60):
    print("{:4d} {:10s} {}".format(n, name, where[:150]))
