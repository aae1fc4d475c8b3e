# r05ze: single-pass cLN (look-back chain): parity, A/B against the three-launch form at the causal TCN shapes, causal bench, golden tests
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "cln" 2>&1 | tail -3 )
cat > /tmp/cln_ab.py <<'P'
import torch, sepkernels, time
K = sepkernels.HipBackend()
dev = "cuda"
for (B, C, T, ldt) in [(16, 512, 3999, 4096), (16, 128, 3999, 4096), (4, 64, 31999, 32000)]:
    x = torch.randn(B, C, ldt, device=dev); dy = torch.randn(B, C, ldt, device=dev)
    g = torch.randn(C, device=dev); b = torch.randn(C, device=dev); al = torch.tensor([0.25], device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    mean = torch.empty(B, ldt, device=dev); rstd = torch.empty(B, ldt, device=dev)
    ws = torch.empty((K.cln_ws_bytes(B, C, T, ldt) + 7) // 8, device=dev, dtype=torch.float64)
    pg, pb, pa = (torch.empty(B, C, device=dev) for _ in range(3))
    def run(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
    tf = run(lambda: K.cln_fwd(x, g, b, y, mean, rstd, ws, B, C, T, ldt, 1e-12, alpha=al))
    tb = run(lambda: K.cln_bwd(dy, x, g, mean, rstd, dx, pg, pb, ws, B, C, T, ldt, 1e-12, alpha=al, dalpha_part=pa))
    nbytes = B * C * ldt * 4
    print("B%d C%d T%d  fwd %.1f us (%.2f TB/s of 2x)  bwd %.1f us (%.2f TB/s of 3x)" % (B, C, T, tf, 2 * nbytes / tf / 1e6, tb, 3 * nbytes / tb / 1e6))
P
for ch in 1 0; do echo "== SEPK_CLN_CHAIN=$ch"; SEPK_CLN_CHAIN=$ch timeout 300 python /tmp/cln_ab.py; done
true
for ch in 1; do SEPK_CLN_CHAIN=$ch timeout 300 python bench.py --config causal --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05ze_causal_$ch.json; python -c "
import json; d=json.load(open('gpurun_out/r05ze_causal_$ch.json')); print('causal chain=$ch', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"; done
