# r08i (experiment, not in the tree): the depthwise backward forms gLN2's sums itself, the sample's 512 rows meet in the middle of the kernel
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
run() { env $1 timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>gpurun_out/r08i_err.txt | tail -n 1 > gpurun_out/r08i_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/r08i_tmp.json')); print('$1', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08i_summary.txt; }
for rep in 1 2; do
  run SEPK_DWB_MEET=0
  run SEPK_DWB_MEET=1
done
SEPK_DWB_MEET=1 timeout 200 python bench.py --eager --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r08i_kt.json
python - <<'PY' | tee -a gpurun_out/r08i_summary.txt
import json
d=json.load(open('profiles/bench_detail.json'))
bk=d['roofline_by_kernel']
for k in ('depthwise bwd','depthwise fwd','gemm heads^T'):
    v=bk[k]; print(k, round(v['avg_us'],1))
PY
