# r04b: raw G rows of the next-but-one chunk read as soon as a row block is split (no LDS round trip in front of the first split)
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "wgrad or golden or oracle" 2>&1 | tail -3 ) > gpurun_out/r04b_gputests.txt; cat gpurun_out/r04b_gputests.txt
python tools/gemm_bench.py --only W2,W3,W4 --reps 30 2>&1 | grep "^W" | tee gpurun_out/r04b_wgrad_bench.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step  wgrad", round(1e3*d["roofline_wgrad"]["avg_launch_ms"],1), "wg-heads", round(k["wgrad heads"]["avg_us"],1), "wg-conv1", round(k["wgrad conv1"]["avg_us"],1), "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" before)
  python bench.py $B 2>gpurun_out/r04b_new.err | python -c "$summ" raw-read-early
done
} > gpurun_out/r04b_ab.txt 2>&1; cat gpurun_out/r04b_ab.txt; tail -3 gpurun_out/r04b_new.err
