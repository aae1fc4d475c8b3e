"""Manual measurement, kept under tests/ because it runs the oracle port (documentation number only, SURVEY.md section 8d "hipified baseline"): the same Conv-TasNet step written
with stock torch.nn.functional ops (oracle/fast_port.py) running on the GPU through MIOpen/rocBLAS, paper-best B=16, so that
the gain of the hand-written path is not confused with the gain of the device."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
from oracle import fast_port as FP  # noqa: E402
from models.conv_tasnet import ConvTasNet  # noqa: E402
from bench import PAPER, T_SAMPLES  # noqa: E402

torch.manual_seed(111)
model = ConvTasNet(**PAPER)
cfg = model.get_config()
p = {k: v.detach().cuda().requires_grad_(True) for k, v in model.state_dict().items()}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
src = (0.1 * torch.randn(B, 2, T_SAMPLES)).cuda()
mix = src.sum(1, keepdim=True)
for _ in range(2):
    FP.train_step(p, cfg, mix, src)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    FP.train_step(p, cfg, mix, src)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("stock PyTorch-ROCm (MIOpen/rocBLAS) fwd+PIT+bwd, B={}: {:.1f} ms/step = {:.0f} frames/s".format(B, 1e3 * dt, B * 3999 / dt))
