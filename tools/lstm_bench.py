"""Development tool: the DPRNN bi-LSTM (hidden 128, 64 features) through libsepkernels vs torch.nn.LSTM (MIOpen), fwd+bwd,
at the two sequence shapes of DPRNN-TasNet config 4 (B=2: intra 510 x 250, inter 500 x 255) and at B=8."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
from sepkernels.functional import lstm_bidirectional  # noqa: E402


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rnn = torch.nn.LSTM(64, 128, batch_first=True, bidirectional=True).cuda()
for nseq, L in ((510, 250), (500, 255), (2040, 250)):
    x = torch.randn(nseq, L, 64, device="cuda", requires_grad=True)

    def ours():
        y = lstm_bidirectional(x, rnn)
        y.sum().backward()

    def miopen():
        y, _ = rnn(x)
        y.sum().backward()
    a, b = timeit(ours), timeit(miopen)
    y1 = lstm_bidirectional(x, rnn)
    y2, _ = rnn(x)
    print("nseq {:5d} L {:3d}: libsepkernels {:8.2f} ms   nn.LSTM(MIOpen) {:8.2f} ms   max|diff| {:.2e}".format(nseq, L, a, b, (y1 - y2).abs().max().item()))
