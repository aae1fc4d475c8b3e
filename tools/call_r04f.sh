# r04f: depthwise backward at dilations 1 and 2: neighbours by two float4 LDS reads + register selects instead of eight scalar reads
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "dwconv or golden" 2>&1 | tail -3 ) > gpurun_out/r04f_gputests.txt; cat gpurun_out/r04f_gputests.txt
python tools/stream_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04f_dwbwd.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step dwbwd", round(k["depthwise bwd"]["avg_us"],1))'
for rep in 1 2; do python bench.py $B 2>/dev/null | python -c "$summ" new; done | tee gpurun_out/r04f_ab.txt
