# r05zy: dense layers in the two-part fp16 arithmetic (linear16_kernel): parity, timing against the fp32-MFMA form at the models' shapes, benches
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear or lstm" 2>&1 | tail -2 )
timeout 300 python /dev/stdin <<'P'
import torch, sepkernels
K = sepkernels.HipBackend(); dev = "cuda"
def run(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
shapes = {"sepformer ff1 33k x 256 -> 1024": (33000, 256, 1024), "sepformer ff2 33k x 1024 -> 256": (33000, 1024, 256), "sepformer qkv 33k x 256 -> 768": (33000, 256, 768),
          "dprnn gates 128k x 64 -> 1024": (128500, 64, 1024), "dprnn fc 128k x 256 -> 64": (128500, 256, 64), "dptnet qkv 64k x 64 -> 192": (64250, 64, 192)}
for name, (ntok, Kin, N) in shapes.items():
    x = torch.randn(ntok, Kin, device=dev); w = torch.randn(N, Kin, device=dev) * Kin ** -0.5; b = torch.randn(N, device=dev)
    y = torch.empty(ntok, N, device=dev); dy = torch.randn(ntok, N, device=dev); dx = torch.empty(ntok, Kin, device=dev)
    ns = max(1, min(512 // max(1, (N // (128 if N % 128 == 0 else 64)) * (Kin // (128 if Kin % 128 == 0 else 64))), (ntok + 255) // 256))
    part = torch.empty(ns, N, Kin, device=dev); pb = torch.empty(ns, N, device=dev)
    row = []
    for ar in ("f32", "f16x3"):
        sepkernels.set_gemm_arith(ar)
        tf = run(lambda: K.linear_fwd(x, w, b, None, y, ntok, Kin, N))
        ti = run(lambda: K.linear_bwd_input(dy, w, dx, ntok, Kin, N, 0))
        tw = run(lambda: K.linear_bwd_weight(dy, x, Kin, part, pb, ntok, Kin, N, 1, 0, ns))
        row.append((tf, ti, tw))
        if ar == "f16x3":
            ref = (x.double() @ w.double().t() + b.double())
            err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    print("%-34s f32 fwd %6.1f din %6.1f dw %6.1f | f16x3 fwd %6.1f din %6.1f dw %6.1f us  (fwd err %.1e)" % ((name,) + row[0] + row[1] + (err,)))
P
( timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "sibling or dprnn or dptnet" 2>&1 | tail -2 )
for c in sepformer dptnet dprnn galrnet; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zy_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r05zy_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
done
