# r03c: gLN2's backward sums from the heads' weight gradient (heads^T without row-sum epilogue / z read): tests, A/B against 6b561e6 on one box, kernel trace
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03c_gputests.txt; cat gpurun_out/r03c_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" prev)
  python bench.py $B 2>/dev/null | python -c "$summ" new
done
} > gpurun_out/r03c_ab.txt; cat gpurun_out/r03c_ab.txt
bash tools/profile_step.sh r03c 6 2>&1 | tail -3
head -40 gpurun_out/r03c_kernel_stats.md
