"""Experiment: the utterances of a batch are independent all the way through Conv-TasNet (gLN is per sample), so a batch of 16 can run as
n independent sub-batches on n HIP streams.  Every kernel of the step fills the chip with ONE wave of workgroups that move through
prologue / main loop / epilogue in lock step; two half-size launches from two streams share each CU with their phases decorrelated.
Times forward + PIT + backward (no optimiser) for n = 1, 2, 4 with the weight gradients on the main stream (SEPK_SIDE_STREAM=0) and
on the side stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import sepkernels  # noqa: E402
from bench_legs import PAPER, T_SAMPLES  # noqa: E402
from models.conv_tasnet import ConvTasNet  # noqa: E402
from criterion.sdr import NegSISDR  # noqa: E402
from criterion.pit import PIT1d  # noqa: E402

sepkernels.load()
dev = torch.device("cuda", 0)
torch.manual_seed(111)
model = ConvTasNet(**PAPER).to(dev)
crit = PIT1d(NegSISDR(), n_sources=2)
B = 16
src = (0.1 * torch.randn(B, 2, T_SAMPLES, generator=torch.Generator().manual_seed(111))).to(dev)
mix = src.sum(1, keepdim=True).contiguous()
flat = model.flat_parameters()


def run(n, steps=12, warm=4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    sinks = [torch.zeros_like(flat) for _ in range(n)]
    per = B // n
    parts = [(mix[i * per:(i + 1) * per].contiguous(), src[i * per:(i + 1) * per].contiguous()) for i in range(n)]

    def step():
        cur = torch.cuda.current_stream(dev)
        losses = []
        for s in streams:
            s.wait_stream(cur)
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                est = model(parts[i][0])
                losses.append((crit(est, parts[i][1])[0], s))
        for i, (loss, s) in enumerate(losses):
            with torch.cuda.stream(s):
                model._grad_sink = sinks[i]
                for q in model.parameters():
                    q.grad = None
                loss.backward()
        model._grad_sink = None
        for s in streams:
            cur.wait_stream(s)
        g = sinks[0]
        for k in range(1, n):
            g = g + sinks[k]
        return g

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g = step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, g / n


ref = None
for side in ("0", "1"):
    os.environ["SEPK_SIDE_STREAM"] = side
    for n in (1, 2, 4):
        ms, g = run(n)
        if ref is None:
            ref = g
        print("side stream {}  sub-batches {}: {:.2f} ms fwd+PIT+bwd   gradient vs one batch: {:.2e}".format(side, n, ms, ((g - ref).abs().max() / ref.abs().max()).item()), flush=True)
