# r03g: publishers without fences, weight-gradient operands fetched in chunk pairs; cooperative vs producer/consumer kernel on the short-K shapes
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "wgrad or dwconv or golden or oracle or sample_aligned" 2>&1 | tail -5 ) > gpurun_out/r03g_gputests.txt; cat gpurun_out/r03g_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" prev)
  python bench.py $B --no-pmc --no-stock 2>gpurun_out/r03g_new.err | python -c "$summ" new
  SEPK_WPC_NS=2 python bench.py $B --no-pmc --no-stock 2>/dev/null | python -c "$summ" new-ns2
done
} > gpurun_out/r03g_ab.txt 2>&1; cat gpurun_out/r03g_ab.txt; tail -3 gpurun_out/r03g_new.err
bash tools/profile_step.sh r03g 6 2>&1 | tail -2
head -30 gpurun_out/r03g_kernel_stats.md
{
echo "== default (coop for K < 512)"; python tools/gemm_bench.py --packed --only F2,G3p,G1 --reps 30
echo "== SEPK_PC_MINK=128"; SEPK_PC_MINK=128 python tools/gemm_bench.py --packed --only F2,G3p,G1 --reps 30
echo "== wgrad NS 4 / 2"; python tools/gemm_bench.py --only W2,W3 --reps 30; SEPK_WPC_NS=2 python tools/gemm_bench.py --only W2,W3 --reps 30
} > gpurun_out/r03g_gemm_bench.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03g_gemm_bench.txt
