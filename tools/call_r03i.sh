# r03i: what the depthwise backward's sums cost; streams / graph; weight-gradient launch anatomy
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
{
for m in none sums publish; do echo "== DWB_MODE=$m"; DWB_MODE=$m python tools/stream_bench.py | tail -3; done
} > gpurun_out/r03i_dwbwd.txt 2>&1; grep -v amdgpu gpurun_out/r03i_dwbwd.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock --no-kernel-timing"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step", d["config"]["launch"])'
{
for rep in 1 2; do
  python bench.py $B 2>/dev/null | python -c "$summ" two-streams
  SEPK_SIDE_STREAM=0 python bench.py $B 2>/dev/null | python -c "$summ" one-stream
  SEPK_GRAPH=1 python bench.py $B 2>/dev/null | python -c "$summ" graph-two-streams
  SEPK_GRAPH=1 SEPK_SIDE_STREAM=0 python bench.py $B 2>/dev/null | python -c "$summ" graph-one-stream
done
} > gpurun_out/r03i_streams.txt 2>&1; cat gpurun_out/r03i_streams.txt
SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels_wpcprof.so SEPK_WPC_NS=2 python tools/wpc_prof.py 2>&1 | tail -3
