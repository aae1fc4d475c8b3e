# r03h: gLN1 means on the consumer side again (sums-only depthwise backward); 512-row cooperative tiles (SEPK_COOP_MI=4); raw-ring 2 vs pairs
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "gemm or dwconv or golden or oracle" 2>&1 | tail -5 ) > gpurun_out/r03h_gputests.txt; cat gpurun_out/r03h_gputests.txt
( SEPK_COOP_MI=4 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "conv1_shape or two_sources_plain or model_shapes or oracle" 2>&1 | tail -5 ) > gpurun_out/r03h_gputests_mi4.txt; cat gpurun_out/r03h_gputests_mi4.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" prev)
  python bench.py $B --no-pmc --no-stock 2>gpurun_out/r03h_new.err | python -c "$summ" new
  SEPK_COOP_MI=4 python bench.py $B --no-pmc --no-stock 2>/dev/null | python -c "$summ" new-mi4
  SEPK_COOP_MI=4 SEPK_WPC_NS=2 python bench.py $B --no-pmc --no-stock 2>/dev/null | python -c "$summ" new-mi4-ns2
done
} > gpurun_out/r03h_ab.txt 2>&1; cat gpurun_out/r03h_ab.txt; tail -3 gpurun_out/r03h_new.err
SEPK_COOP_MI=4 bash tools/profile_step.sh r03h 6 2>&1 | tail -2
head -24 gpurun_out/r03h_kernel_stats.md
