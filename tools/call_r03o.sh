# r03o: depthwise^T re-forming z from `a` (3 HBM streams instead of 4); non-temporal hint on the GEMMs' activation DMA; live PMC traffic
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "dwconv or golden or gemm or oracle" 2>&1 | tail -3 ) > gpurun_out/r03o_gputests.txt; cat gpurun_out/r03o_gputests.txt
{ echo "== recompute z"; python tools/stream_bench.py 2>&1 | grep -v amdgpu; echo "== read z"; DWB_READ_Z=1 python tools/stream_bench.py 2>&1 | grep -v amdgpu; } > gpurun_out/r03o_dwbwd.txt; cat gpurun_out/r03o_dwbwd.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step  gemm", round(1e3*d["roofline"]["avg_launch_ms"],1), round(d["roofline"]["frac"],3), " wgrad", round(1e3*d["roofline_wgrad"]["avg_launch_ms"],1), " dwbwd", round(k["depthwise bwd"]["avg_us"],1), "heads", round(k["gemm heads"]["avg_us"],1), "conv1T", round(k["gemm conv1^T"]["avg_us"],1), "headsT", round(k["gemm heads^T"]["avg_us"],1), "conv1", round(k["gemm conv1"]["avg_us"],1), "loss", d["config"]["final_loss"])'
NT=$PWD/dnn-based_source_separation_amd/libsepkernels_nt.so
{
for rep in 1 2; do
  SEPK_DWB_RECOMPUTE=0 python bench.py $B 2>/dev/null | python -c "$summ" read-z
  python bench.py $B 2>gpurun_out/r03o_new.err | python -c "$summ" recompute-z
  SEPKERNELS_LIB=$NT python bench.py $B 2>/dev/null | python -c "$summ" recompute-z+nt
done
} > gpurun_out/r03o_ab.txt 2>&1; cat gpurun_out/r03o_ab.txt; tail -3 gpurun_out/r03o_new.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-stock > gpurun_out/r03o_bench.json 2> gpurun_out/r03o_bench.err; tail -2 gpurun_out/r03o_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03o_bench.json').read().strip().splitlines()[-1])
h=d.get('hbm_traffic') or {}
print('step', d['ms_per_step'], 'traffic GB', h.get('step_total_GB'), 'over', h.get('over_algorithmic'))
pk=h.get('per_kernel',{})
rows=sorted(((v['launches_per_step']*(v['read_MB']+v['write_MB'])/1e3,k[:60],v['launches_per_step'],v['read_MB'],v['write_MB']) for k,v in pk.items()),reverse=True)[:14]
for r in rows: print("%.2f GB  %-60s %5.1f r %.0f w %.0f"%r)
PY
