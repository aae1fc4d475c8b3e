# r05zk: gLN over tokens in slices for few long sequences (GALRNet): parity, bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gln_tokens" 2>&1 | tail -2 )
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "sibling" 2>&1 | tail -2 )
for c in galrnet dptnet; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zk_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r05zk_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
done
