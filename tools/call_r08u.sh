cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for v in 0 1; do
SEPK_WGRAD_PRESPLIT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r08u_$v.json
python - <<PY | tee -a gpurun_out/r08u_summary.txt
import json
l=json.load(open('gpurun_out/r08u_$v.json')); print('presplit $v step', l['ms_per_step'])
d=json.load(open('profiles/bench_detail.json'))
bk=d['roofline_by_kernel']
for k in ('wgrad heads','gln sums from wgrad','gemm heads^T','depthwise bwd','gemm conv1^T','split_rows'):
    if k in bk: print('  ', k, round(bk[k]['avg_us'],1), bk[k]['launches_per_step'])
PY
done
