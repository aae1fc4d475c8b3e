# r05s: staged causal path with the two heads of a layer as one product (PaddedHeadsFn): parity + bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "golden_forward_loss_grads" 2>&1 | tail -2 ) | tee gpurun_out/r05s_model.txt
timeout 300 python bench.py --config causal --steps 6 --warmup 2 2>/dev/null | tail -n 1 > gpurun_out/r05s_causal.json; python -c "
import json; d=json.load(open('gpurun_out/r05s_causal.json')); print('causal staged B=16', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'mem', round(d['peak_memory_GB'],1))"
