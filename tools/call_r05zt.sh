# r05zt: attention timing after the exp2-domain change (compare profiles/r05zo_attention.txt) + SepFormer bench twice
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 300 python /dev/stdin <<'P'
import torch, sepkernels
K = sepkernels.HipBackend(); dev = "cuda"
def run(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, (N, H, L, D, pd) in {"dptnet intra": (257, 4, 250, 16, 0.0), "sepformer": (132, 8, 250, 32, 0.1), "sepformer p0": (132, 8, 250, 32, 0.0), "galr": (128, 8, 81, 8, 0.0)}.items():
    qkv = torch.randn(N, L, 3, H, D, device=dev); dout = torch.randn(N, L, H, D, device=dev)
    o = torch.empty(N, L, H, D, device=dev); lse = torch.empty(N, H, L, device=dev); delta = torch.empty_like(lse); dq = torch.empty_like(qkv)
    tf = run(lambda: K.attn_fwd(qkv, o, lse, N, L, H, D, D ** -0.5, pd, 77))
    tb = run(lambda: K.attn_bwd(qkv, o, dout, lse, delta, dq, N, L, H, D, D ** -0.5, pd, 77))
    print("%-13s attn.hip fwd %6.1f bwd %6.1f us" % (name, tf, tb))
P
for i in 1 2; do timeout 300 python bench.py --config sepformer --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zt_sepformer.json; python -c "
import json; d=json.load(open('gpurun_out/r05zt_sepformer.json')); print('sepformer', round(d['ms_per_step'],2), 'ms')"; done
