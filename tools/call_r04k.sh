# r04k: stand-alone gLN statistics with a grid-stride loop (DPRNN-TasNet's long rows)
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "gln or dprnn or sibling" 2>&1 | tail -3 ) | tee gpurun_out/r04k_gputests.txt
summ2='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), "ms/step loss", d["config"]["final_loss"])'
for cfg in dprnn dprnn galrnet; do python bench.py --config $cfg --steps 8 --warmup 3 2>/dev/null | python -c "$summ2" $cfg; done | tee gpurun_out/r04k_dual.txt
