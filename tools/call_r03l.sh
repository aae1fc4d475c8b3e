export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/w16_probe tools/w16_probe.hip 2>/dev/null && /tmp/w16_probe
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "wgrad or golden or oracle" 2>&1 | tail -5 ) > gpurun_out/r03l_gputests.txt; cat gpurun_out/r03l_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  SEPK_WGRAD_F16=0 python bench.py $B 2>/dev/null | python -c "$summ" wgrad-bf16x6
  python bench.py $B 2>gpurun_out/r03l_new.err | python -c "$summ" wgrad-f16x3
done
} > gpurun_out/r03l_ab.txt 2>&1; cat gpurun_out/r03l_ab.txt; tail -3 gpurun_out/r03l_new.err
python tools/gemm_bench.py --only W2,W3 --reps 30 2>&1 | grep -v amdgpu
