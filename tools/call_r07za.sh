# r07za: four / two heads of a short sequence per attention workgroup
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
timeout 300 python tools/attn_bench.py sepformer-inter sepformer-intra galrnet | tee gpurun_out/r07za_attention.txt
timeout 300 python tools/gpu_fuzz_round5.py 60 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "sibling or sepformer" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for c in sepformer; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07za_bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r07za_bench_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'roofline', d['roofline']['bound'], round(d['roofline']['achieved'],1), d['roofline']['unit'], round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
