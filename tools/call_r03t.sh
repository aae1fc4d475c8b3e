# r03t: f16x3 weight gradient with the producers one chunk ahead (three operand buffers); BLAS library choice for the DPRNN-TasNet step
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "wgrad or golden or oracle" 2>&1 | tail -3 ) > gpurun_out/r03t_gputests.txt; cat gpurun_out/r03t_gputests.txt
{ python tools/gemm_bench.py --only W1,W2,W3,W4 --reps 30 2>&1 | grep "^W"
  SEPKERNELS_LIB=$R/dnn-based_source_separation_amd/libsepkernels_wpcprof.so python tools/wpc16_prof.py 2>&1 | grep -v amdgpu; } > gpurun_out/r03t_wpc16_stamps.txt; cat gpurun_out/r03t_wpc16_stamps.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step  wgrad", round(1e3*d["roofline_wgrad"]["avg_launch_ms"],1), "wg-heads", round(k["wgrad heads"]["avg_us"],1), "wg-conv1", round(k["wgrad conv1"]["avg_us"],1), "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" operands-at-barrier)
  python bench.py $B 2>gpurun_out/r03t_new.err | python -c "$summ" one-chunk-ahead
done
} > gpurun_out/r03t_ab.txt 2>&1; cat gpurun_out/r03t_ab.txt; tail -3 gpurun_out/r03t_new.err
summ2='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), "ms/step loss", d["config"]["final_loss"])'
{
python bench.py --config dprnn --steps 8 --warmup 3 2>/dev/null | python -c "$summ2" dprnn-default-blas
TORCH_BLAS_PREFER_HIPBLASLT=0 python bench.py --config dprnn --steps 8 --warmup 3 2>/dev/null | python -c "$summ2" dprnn-rocblas
TORCH_BLAS_PREFER_HIPBLASLT=1 python bench.py --config dprnn --steps 8 --warmup 3 2>/dev/null | python -c "$summ2" dprnn-hipblaslt
} > gpurun_out/r03t_dprnn_blas.txt 2>&1; cat gpurun_out/r03t_dprnn_blas.txt
