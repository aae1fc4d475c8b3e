"""Development tool: which torch additions does a training step of the staged causal Conv-TasNet issue?  (shapes and counts of aten::add / add_ / sum ...)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from torch.profiler import profile, ProfilerActivity
from models.conv_tasnet import ConvTasNet
from criterion.pit import PIT1d
from criterion.sdr import NegSISDR
PAPER = dict(n_basis=512, kernel_size=16, stride=8, enc_basis="trainable", dec_basis="trainable", enc_nonlinear=None, sep_hidden_channels=512,
             sep_bottleneck_channels=128, sep_skip_channels=128, sep_kernel_size=3, sep_num_blocks=3, sep_num_layers=8, dilated=True, separable=True,
             sep_nonlinear="prelu", sep_norm=True, mask_nonlinear="sigmoid", n_sources=2)
torch.manual_seed(0)
model = ConvTasNet(**dict(PAPER, causal=True)).cuda()
assert model.staged, model.staged_reason
crit = PIT1d(NegSISDR(), n_sources=2)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
src = 0.1 * torch.randn(16, 2, 32000, device="cuda"); mix = src.sum(1, keepdim=True).contiguous()
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = crit(model(mix), src)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_time_total > 0 and e.name.startswith("aten::"):
        k = (e.name, str(e.input_shapes)[:90])
        agg[k][0] += 1; agg[k][1] += e.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]
for (name, shp), (n, t) in rows:
    print("%-28s %4d  %8.1f us  %s" % (name, n, t, shp))
