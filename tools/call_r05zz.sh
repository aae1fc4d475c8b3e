# r05zz: SepFormer's feed-forward pairs on the 1x1-convolution kernels: device check against torch's own layer, bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 300 python /dev/stdin <<'P'
import torch, sys
sys.path.insert(0, "dnn-based_source_separation_amd/src")
from models.sepformer import _ChunkPathEncoder, _ff_on_conv_kernels
torch.manual_seed(3)
for (N, L, C, Fd, train) in [(5, 250, 256, 1024, False), (3, 100, 128, 256, True)]:
    layer = torch.nn.TransformerEncoderLayer(C, 8, Fd, dropout=0.0, activation="relu", batch_first=False).cuda().train(train)
    x = torch.randn(N, L, C, device="cuda", requires_grad=True)
    assert _ff_on_conv_kernels(layer, x)
    y = _ChunkPathEncoder._layer_tokens(layer, x)
    l64 = torch.nn.TransformerEncoderLayer(C, 8, Fd, dropout=0.0, activation="relu", batch_first=False).double().train(train)
    l64.load_state_dict({k: v.double().cpu() for k, v in layer.state_dict().items()})
    xr = x.detach().double().cpu().requires_grad_(True)
    ref = l64(xr.transpose(0, 1)).transpose(0, 1)
    w = torch.randn_like(y)
    g = torch.autograd.grad((y * w).sum(), [x] + list(layer.parameters()))
    r = torch.autograd.grad((ref * w.double().cpu()).sum(), [xr] + list(l64.parameters()))
    eo = ((y.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    eg = max(((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30)).item() for a, b in zip(g, r))
    print("layer %d x %d -> %d: output err %.1e, worst gradient err %.1e" % (N * L, C, Fd, eo, eg))
    assert eo < 1e-4 and eg < 1e-3
P
for i in 1 2; do timeout 300 python bench.py --config sepformer --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zz_sepformer.json; python -c "
import json; d=json.load(open('gpurun_out/r05zz_sepformer.json')); print('sepformer', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"; done
