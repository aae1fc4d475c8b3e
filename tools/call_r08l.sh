# r08l: where the fp32-MFMA pass spends its time (SEPK_GEMM_ARITH=f32, HIP events per launch class)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
SEPK_GEMM_ARITH=f32 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r08l_f32.json
python - <<'PY' | tee gpurun_out/r08l_summary.txt
import json
l=json.load(open('gpurun_out/r08l_f32.json')); print('f32 step', l['ms_per_step'], l['config']['launch'])
d=json.load(open('profiles/bench_detail.json'))
bk=d['roofline_by_kernel']
for k,v in sorted(bk.items(), key=lambda kv:-kv[1]['ms_per_step'])[:16]:
    print(f"{k:45s} n={v['launches_per_step']:5.1f} avg={v['avg_us']:7.1f}us ms={v['ms_per_step']:6.3f} mfma_frac={v.get('matrix_pipe_frac')}")
PY
