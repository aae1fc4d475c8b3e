"""Development tool: DPRNN-TasNet (BASELINE.json configs[3]: N=64, L=2, F=64, H=128, K=250, P=125, 6 blocks, 2 spk, 4 s @ 8 kHz)
forward + PIT(NegSI-SDR) + backward time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
from criterion.pit import PIT1d  # noqa: E402
from criterion.sdr import NegSISDR  # noqa: E402
from models.dprnn_tasnet import DPRNNTasNet  # noqa: E402

torch.manual_seed(111)
model = DPRNNTasNet(n_basis=64, kernel_size=2, stride=1, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None,
                    sep_hidden_channels=128, sep_bottleneck_channels=64, sep_chunk_size=250, sep_hop_size=125, sep_num_blocks=6,
                    sep_norm=True, mask_nonlinear="sigmoid", causal=False, rnn_type="lstm", n_sources=2).cuda()
crit = PIT1d(NegSISDR(), n_sources=2)
for B in (2, 8):
    src = (0.1 * torch.randn(B, 2, 32000)).cuda()
    mix = src.sum(1, keepdim=True)

    def step():
        for p in model.parameters():
            p.grad = None
        loss, _ = crit(model(mix), src)
        loss.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("DPRNN-TasNet config 4, B={}: {:.1f} ms/step ({:.1f} utterances/s)".format(B, 1e3 * dt, B / dt))
