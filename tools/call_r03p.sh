# r03p: row stride of the activation workspaces: 4096 floats (16 KiB, the rows of a column fall into few L2 sets) vs 4352
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
{
for l in 4096 4352; do echo "== ldt $l"; python tools/gemm_bench.py --packed --reps 30 --ldt $l 2>&1 | grep -v amdgpu; LDT=$l python tools/stream_bench.py 2>&1 | grep "per 8"; done
} > gpurun_out/r03p_ldt_kernels.txt 2>&1; cat gpurun_out/r03p_ldt_kernels.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step  gemm", round(1e3*d["roofline"]["avg_launch_ms"],1), round(d["roofline"]["frac"],3), " wgrad", round(1e3*d["roofline_wgrad"]["avg_launch_ms"],1), " dwbwd", round(k["depthwise bwd"]["avg_us"],1), "dwfwd", round(k["depthwise fwd"]["avg_us"],1), "heads", round(k["gemm heads"]["avg_us"],1), "conv1T", round(k["gemm conv1^T"]["avg_us"],1), "headsT", round(k["gemm heads^T"]["avg_us"],1), "conv1", round(k["gemm conv1"]["avg_us"],1), "wg-heads", round(k["wgrad heads"]["avg_us"],1), "wg-conv1", round(k["wgrad conv1"]["avg_us"],1), "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  python bench.py $B 2>/dev/null | python -c "$summ" stride-4096
  SEPK_ROW_PAD=256 python bench.py $B 2>gpurun_out/r03p_new.err | python -c "$summ" stride-4352
done
} > gpurun_out/r03p_ab.txt 2>&1; cat gpurun_out/r03p_ab.txt; tail -3 gpurun_out/r03p_new.err
( SEPK_ROW_PAD=256 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "golden or oracle" 2>&1 | tail -3 ) > gpurun_out/r03p_gputests.txt; cat gpurun_out/r03p_gputests.txt
