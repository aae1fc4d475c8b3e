cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "golden_forward_loss_grads" 2>&1 | tail -2 ) | tee gpurun_out/r05r_model.txt
for basis in fourier pinv; do
timeout 300 python bench.py --basis $basis --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1 > gpurun_out/r05r_$basis.json; python -c "
import json; d=json.load(open('gpurun_out/r05r_$basis.json')); print('$basis', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s loss', d['config']['final_loss'])"; done
python - <<'PY'
import torch, time
w = torch.randn(1, 512, 16, device='cuda', requires_grad=True)
for f, name in ((torch.pinverse, 'pinverse'), (lambda a: torch.linalg.solve(a.transpose(1, 2) @ a, a.transpose(1, 2)), 'normal equations')):
    for _ in range(3): y = f(w); y.sum().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = f(w); y.sum().backward()
    torch.cuda.synchronize(); print(name, 'fwd+bwd', round(1e3 * (time.perf_counter() - t0) / 10, 3), 'ms')
PY
