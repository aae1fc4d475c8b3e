# r04e: dual-path separators without layout copies (tiled transposes, interleaved bi-LSTM output)
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "lstm or linear or chunk" 2>&1 | tail -3 ) > gpurun_out/r04e_gputests.txt; cat gpurun_out/r04e_gputests.txt
( timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_recipe.py -m gpu -x -q -k "dprnn or galr or dpt or sepformer or lstm or dual or sibling or graphed" 2>&1 | tail -5 ) > gpurun_out/r04e_gputests2.txt; cat gpurun_out/r04e_gputests2.txt
summ2='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), "ms/step loss", d["config"]["final_loss"], (d.get("roofline") or {}).get("frac"))'
{
for cfg in dprnn galrnet dptnet; do python bench.py --config $cfg --steps 8 --warmup 3 2>gpurun_out/r04e_$cfg.err | python -c "$summ2" $cfg; done
} > gpurun_out/r04e_dual.txt 2>&1; cat gpurun_out/r04e_dual.txt
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_dprnn -o dprnn -- python $R/bench.py --config dprnn --steps 6 --warmup 2 > /tmp/dprnn.log 2>&1
  db=$(find /tmp/prof_dprnn -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r04e_dprnn 8 )
head -22 gpurun_out/r04e_dprnn_kernel_stats.md | cut -c1-150
