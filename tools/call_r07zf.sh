# r07zf: dense kernel tile choice by shape (64-row tiles for layers 128 or more wide): tests, timing, the four dual-path benches
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear or lstm or rownorm" 2>&1 | grep -E "passed|failed" | tail -2
timeout 300 python tools/linear_bench.py | tee gpurun_out/r07zf_linear.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "sibling or sepformer or dptnet or dprnn" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
for c in sepformer dprnn dptnet galrnet; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07zf_bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r07zf_bench_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
