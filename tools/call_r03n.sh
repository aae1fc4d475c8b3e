# r03n: batched second stages; DPRNN-TasNet / GALRNet steps eager vs captured in a hipGraph
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "finalize or golden or dprnn or galrnet" 2>&1 | tail -3 )
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step", d["config"].get("launch"), "loss", d["config"]["final_loss"])'
for rep in 1 2; do python bench.py $B --no-kernel-timing 2>/dev/null | python -c "$summ" convtasnet; done
for cfg in dprnn galrnet dptnet; do
  python bench.py --config $cfg --steps 10 --warmup 3 2>/dev/null | python -c "$summ" $cfg-eager
  SEPK_GRAPH=1 python bench.py --config $cfg --steps 10 --warmup 3 2>gpurun_out/r03n_$cfg.err | python -c "$summ" $cfg-graph || tail -3 gpurun_out/r03n_$cfg.err
done
