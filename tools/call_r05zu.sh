# r05zu: the three attention models and the attention kernel tests on the last tree
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "attention or sibling" 2>&1 | tail -2 )
for c in dptnet galrnet sepformer; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zu_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r05zu_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
done
