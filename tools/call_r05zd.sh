# r05zd: SepFormer's transformer stacks on token-major rows with the dense layers on csrc/linear.hip: golden parity, bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "sibling or sepformer" 2>&1 | tail -2 )
timeout 300 python bench.py --config sepformer --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zd_sepformer.json; python -c "
import json; d=json.load(open('gpurun_out/r05zd_sepformer.json')); print('sepformer', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config']['final_loss'])"
