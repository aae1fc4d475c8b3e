# r08f: kernel-choice toggles of the short-contraction products on the current tree (recorded sequence, one box, alternating)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1 > gpurun_out/r08f_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/r08f_tmp.json')); print('$1', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08f_summary.txt; }
for rep in 1 2; do
  run SEPK_NONE=1
  run SEPK_PC_MINK=256
  run SEPK_COOP_MI=4
  run SEPK_COOP_NS=3
  run SEPK_PC_22=1
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_recipe.py -x -q -m gpu -k "sepformer or galr or dptnet or dropout or rownorm or relu_drop or attention or recipe or train" 2>&1 | tail -n 6 | tee -a gpurun_out/r08f_summary.txt
