export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels_wpcprof.so python tools/wpc16_prof.py 2>&1 | grep -v amdgpu
python tools/gemm_bench.py --only W --reps 30 2>&1 | grep -v amdgpu
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "wgrad" 2>&1 | tail -3 )
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
for rep in 1 2; do
  SEPK_WGRAD_F16=0 python bench.py $B 2>/dev/null | python -c "$summ" wgrad-bf16x6
  python bench.py $B 2>/dev/null | python -c "$summ" wgrad-f16x3
done
