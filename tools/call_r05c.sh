# r05c: sep_dwconv_bwd direct with the two-level (per-XCD L2, then published) arrival: kernel tests + A/B of the step
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "dwconv" 2>&1 | tail -4 ) | tee gpurun_out/r05c_kernels.txt
for dmode in 1 0; do
  echo "== SEPK_DWB_DIRECT=$dmode"
  SEPK_DWB_DIRECT=$dmode timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass > gpurun_out/r05c_d${dmode}.out 2> gpurun_out/r05c_d${dmode}.err
  tail -c 300 gpurun_out/r05c_d${dmode}.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05c_d${dmode}.out').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'loss', d['config']['final_loss'])
k=json.load(open('profiles/bench_detail.json'))['roofline_by_kernel']
print({n: round(v['avg_us'],1) for n,v in k.items() if v['share_of_kernel_time']>0.02})
PY
done
python - <<'PY'
import sys; sys.path.insert(0,'dnn-based_source_separation_amd/src')
import sepkernels; sepkernels.load(); K=sepkernels.backend(); print('direct max rows', K.dwconv_bwd_direct_max_rows(4096, True), 'timeouts', K.sync_timeouts())
PY
