# r04d: GraphedStep (the dual-path steps as one hipGraph launch): equality with the eager step, timing beside it
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "graphed" 2>&1 | tail -15 ) > gpurun_out/r04d_gputests.txt; cat gpurun_out/r04d_gputests.txt
summ2='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); g=d.get("graph_replay") or {}; print(sys.argv[1], "eager", round(d["ms_per_step"],2), "ms/step loss", d["config"]["final_loss"], "| graph", g.get("ms_per_step"), g.get("final_loss"), g.get("valid"), g.get("error"))'
for cfg in dprnn galrnet dptnet sepformer; do python bench.py --config $cfg --steps 8 --warmup 3 2>gpurun_out/r04d_$cfg.err | python -c "$summ2" $cfg || tail -3 gpurun_out/r04d_$cfg.err; done 2>&1 | tee gpurun_out/r04d_dual.txt
