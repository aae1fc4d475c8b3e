# r05g: direct with the single-round-trip post hidden under the next row's loads: kernel tests, model parity, A/B (timeouts read in-process)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "dwconv" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300 | head -8 ) | tee gpurun_out/r05g_kernels.txt
for rep in 1 2; do
for dmode in 1 0; do
  echo "== SEPK_DWB_DIRECT=$dmode rep $rep"
  SEPK_DWB_DIRECT=$dmode SEPK_BENCH_TIMEOUTS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass > gpurun_out/r05g_d${dmode}_$rep.out 2> gpurun_out/r05g_d${dmode}_$rep.err
  tail -c 300 gpurun_out/r05g_d${dmode}_$rep.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05g_d${dmode}_$rep.out').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'loss', d['config']['final_loss'], 'timeouts', d['config'].get('sync_timeouts'))
k=json.load(open('profiles/bench_detail.json'))['roofline_by_kernel']
print({n: round(v['avg_us'],1) for n,v in k.items() if v['share_of_kernel_time']>0.02})
PY
done; done
( timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3 ) | tee gpurun_out/r05g_model.txt
