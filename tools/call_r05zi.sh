# r05zi: the staged layers' heads / heads^T on the f16x3 instantiations of the direct kernel (they fell to the fp32-MFMA generic form): tests, causal bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -2 )
( timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "causal or staged or golden" 2>&1 | tail -2 )
timeout 300 python bench.py --config causal --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zi_causal.json; python -c "
import json; d=json.load(open('gpurun_out/r05zi_causal.json')); print('causal', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
