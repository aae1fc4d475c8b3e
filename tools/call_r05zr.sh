# r05zr: reduce_slabs with four slab phases per output: tests, headline and DPTNet benches
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "reduce or wgrad or linear" 2>&1 | tail -2 )
( timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden or oracle or batch16" 2>&1 | tail -2 )
python bench.py --no-cpu-baseline --no-stock --no-pmc 2>/dev/null | tail -n 1 > gpurun_out/r05zr_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r05zr_bench.json')); print('headline', round(d['ms_per_step'],3), 'ms')"
python - <<'P'
import json
d = json.load(open('profiles/bench_detail.json'))
for k, v in (d.get('roofline_by_kernel') or {}).items():
    if 'reduce' in k or 'finalize' in k: print(k, v.get('avg_us'), v.get('ms_per_step'))
P
for c in dptnet dprnn causal; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zr_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r05zr_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
done
