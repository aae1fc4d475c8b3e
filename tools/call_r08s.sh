# r08s: dS pre-split once per step (sep_split_rows + G2_pre) against splitting it in every heads weight gradient; kernel and model tests
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "wgrad or tcn_layer or golden or paper_best or batch16 or recorded" 2>&1 | grep -E "passed|failed|Error|assert" | head -5 | tee gpurun_out/r08s_summary.txt
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1 > gpurun_out/r08s_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/r08s_tmp.json')); print('$1', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08s_summary.txt; }
for rep in 1 2 3; do
  run SEPK_WGRAD_PRESPLIT=0
  run SEPK_WGRAD_PRESPLIT=1
done
