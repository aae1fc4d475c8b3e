# r08d: depthwise backward with ONE LDS row (v1 and dz share it: 6 workgroups per unit instead of 4) against the previous commit, one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
  for tree in _ab_prev .; do
    (cd $tree && PYTHONPATH=dnn-based_source_separation_amd/src timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1) > gpurun_out/r08d_tmp.json
    python -c "
import json; d=json.load(open('gpurun_out/r08d_tmp.json')); print('tree $tree', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08d_summary.txt
  done
done
PYTHONPATH=dnn-based_source_separation_amd/src timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dwconv or tcn_layer" 2>&1 | tail -n 3 | tee -a gpurun_out/r08d_summary.txt
