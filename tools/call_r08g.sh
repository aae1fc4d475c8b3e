# r08g: the 512 x 64 cooperative tile per shape (SEPK_COOP_MI4 bits: 1 conv1, 2 heads^T, 4 plain), one box, alternating
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
run() { env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1 > gpurun_out/r08g_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/r08g_tmp.json')); print('$1', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08g_summary.txt; }
for rep in 1 2 3; do
  run SEPK_COOP_MI4=0
  run SEPK_COOP_MI4=1
  run SEPK_COOP_MI4=2
  run SEPK_COOP_MI4=3
  run SEPK_COOP_MI4=7
done
SEPK_COOP_MI4=7 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "gemm or tcn_layer or golden or paper_best or batch16" 2>&1 | tail -n 4 | tee -a gpurun_out/r08g_summary.txt
