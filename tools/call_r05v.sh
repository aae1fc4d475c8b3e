# r05v: DPTNet on token-major rows (sep_gln_tokens_*, batch-first attention / LSTM / Linear): kernel + golden parity, bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "gln_tokens" 2>&1 | tail -2 )
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "sibling or dptnet" 2>&1 | tail -2 )
timeout 300 python bench.py --config dptnet --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05v_dptnet.json; python -c "
import json; d=json.load(open('gpurun_out/r05v_dptnet.json')); print('dptnet', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config']['final_loss'])"
