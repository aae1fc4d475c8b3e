# r08v: dS pre-split at small batches (up to 64 slabs per sample): tests, B = 2 / 4 with and without
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "presplit or golden or paper_best or recorded" 2>&1 | grep -E "passed|failed" | tee gpurun_out/r08v_summary.txt
run() { env $1 timeout 300 python bench.py --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1 > gpurun_out/r08v_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/r08v_tmp.json')); print('B=$2 $1', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08v_summary.txt; }
for rep in 1 2; do for b in 2 4; do
  run SEPK_WGRAD_PRESPLIT=0 $b
  run SEPK_WGRAD_PRESPLIT=1 $b
done; done
