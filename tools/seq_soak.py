"""Development tool: N training steps of paper-best Conv-TasNet as a recorded launch sequence against the same N eager steps (fresh data every step):
the replayed step must keep training like the eager one (the hipGraph replay it replaced ended in inf / 49.98 losses in half the runs).
    python tools/seq_soak.py [--steps 200] [--batch 4] [--runs 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from bench_legs import PAPER  # noqa: E402
from models.conv_tasnet import ConvTasNet  # noqa: E402
from criterion.sdr import NegSISDR  # noqa: E402
from criterion.pit import PIT1d  # noqa: E402
from sepkernels.train import FusedTrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--runs", type=int, default=3)
args = ap.parse_args()


def run(recorded, seed):
    torch.manual_seed(111)
    model = ConvTasNet(**PAPER).cuda()
    step = FusedTrainStep(model, PIT1d(NegSISDR(), n_sources=2), lr=1e-3, max_norm=5.0, auto_record=recorded)
    g = torch.Generator().manual_seed(seed)
    losses = []
    for _ in range(args.steps):
        src = (0.1 * torch.randn(args.batch, 2, 32000, generator=g)).cuda()
        losses.append(step(src.sum(1, keepdim=True).contiguous(), src).clone())      # (a replayed step returns its recorded loss BUFFER)
    torch.cuda.synchronize()
    return torch.stack([x.reshape(()) for x in losses]).cpu()


for r in range(args.runs):
    le, lr = run(False, 5 + r), run(True, 5 + r)
    dev = ((le - lr).abs() / le.abs().clamp_min(1e-3))
    print("run {}: {} steps, B = {}: eager first / last loss {:.4f} / {:.4f}, recorded {:.4f} / {:.4f}; largest relative difference over the first 20 steps "
          "{:.2e}, over all {:.2e}; all finite: {}".format(r, args.steps, args.batch, le[0].item(), le[-1].item(), lr[0].item(), lr[-1].item(),
                                                         dev[:20].max().item(), dev.max().item(), bool(torch.isfinite(lr).all())), flush=True)
