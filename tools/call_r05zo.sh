# r05zo: attention core on csrc/attn.hip: parity, timing against torch's memory-efficient SDPA at the models' shapes, benches
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -3 )
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "sibling" 2>&1 | tail -2 )
timeout 300 python /dev/stdin <<'P'
import torch, sepkernels
import torch.nn.functional as F
K = sepkernels.HipBackend(); dev = "cuda"
def run(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, (N, H, L, D, pd) in {"dptnet intra": (257, 4, 250, 16, 0.0), "dptnet inter": (250, 4, 257, 16, 0.0), "sepformer": (132, 8, 250, 32, 0.1), "galr": (128, 8, 81, 8, 0.0)}.items():
    qkv = torch.randn(N, L, 3, H, D, device=dev); dout = torch.randn(N, L, H, D, device=dev)
    o = torch.empty(N, L, H, D, device=dev); lse = torch.empty(N, H, L, device=dev); delta = torch.empty_like(lse); dq = torch.empty_like(qkv)
    tf = run(lambda: K.attn_fwd(qkv, o, lse, N, L, H, D, D ** -0.5, pd, 77))
    tb = run(lambda: K.attn_bwd(qkv, o, dout, lse, delta, dq, N, L, H, D, D ** -0.5, pd, 77))
    q, k, v = (qkv[:, :, i].transpose(1, 2).contiguous().requires_grad_(True) for i in range(3))
    go = dout.transpose(1, 2).contiguous()
    sf = run(lambda: F.scaled_dot_product_attention(q, k, v, dropout_p=pd))
    def fb():
        oo = F.scaled_dot_product_attention(q, k, v, dropout_p=pd); oo.backward(go); q.grad = k.grad = v.grad = None
    sfb = run(fb)
    print("%-13s attn.hip fwd %6.1f bwd %6.1f us   | torch SDPA fwd %6.1f fwd+bwd %6.1f us" % (name, tf, tb, sf, sfb))
P
for c in dptnet sepformer galrnet; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zo_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r05zo_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
done
