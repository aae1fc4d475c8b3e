# r04h: batched gLN finalize with several rows per wave
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "finalize or golden or oracle" 2>&1 | tail -3 ) > gpurun_out/r04h_gputests.txt; cat gpurun_out/r04h_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step finalize", round(k["gln_bwd_finalize_batch"]["avg_us"],1), "reduce_slabs", round(k["reduce_slabs"]["ms_per_step"],3), "loss", d["config"]["final_loss"])'
for rep in 1 2; do python bench.py $B 2>/dev/null | python -c "$summ" new; done | tee gpurun_out/r04h_ab.txt
