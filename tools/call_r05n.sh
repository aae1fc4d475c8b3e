# r05n: the staged causal path: kernel + model parity, then its bench (and the torch composition of the same model for comparison)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "cln" 2>&1 | tail -3 ) | tee gpurun_out/r05n_kernels.txt
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "golden_forward_loss_grads" 2>&1 | tail -3 ) | tee gpurun_out/r05n_model.txt
timeout 300 python bench.py --config causal --steps 6 --warmup 2 2>/dev/null | tail -n 1 > gpurun_out/r05n_causal.json; python -c "
import json; d=json.load(open('gpurun_out/r05n_causal.json')); print('causal staged B=16', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'mem', round(d['peak_memory_GB'],1))"
timeout 300 python - <<'PY'
import sys, time, torch
sys.path.insert(0, 'tools'); sys.path.insert(0, 'dnn-based_source_separation_amd/src')
from bench_legs import PAPER, T_SAMPLES
from models.conv_tasnet import ConvTasNet
from criterion.sdr import NegSISDR
from criterion.pit import PIT1d
torch.manual_seed(111)
m = ConvTasNet(**dict(PAPER, causal=True)).cuda(); m.staged = False        # the module-by-module torch composition (round 3's path for causal)
crit = PIT1d(NegSISDR(), n_sources=2)
B = 4
src = (0.1 * torch.randn(B, 2, T_SAMPLES, generator=torch.Generator().manual_seed(1))).cuda(); mix = src.sum(1, keepdim=True)
def step():
    for q in m.parameters(): q.grad = None
    crit(m(mix), src)[0].backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print('causal torch composition B=%d: %.1f ms fwd+bwd = %.0f frames/s' % (B, 1e3 * dt, B * 3999 / dt))
PY
