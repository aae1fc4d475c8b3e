# r05p: staged causal path after: float4 row kernels for the TCN depthwise geometry, vector loads of the per-frame cLN statistics, one
# operand bound per pass
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "cln or depthwise" 2>&1 | tail -3 ) | tee gpurun_out/r05p_kernels.txt
( timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_recipe.py -x -q -k "golden_forward_loss_grads or sibling or recipe or dprnn" 2>&1 | tail -3 ) | tee gpurun_out/r05p_model.txt
timeout 300 python bench.py --config causal --steps 6 --warmup 2 2>/dev/null | tail -n 1 > gpurun_out/r05p_causal.json; python -c "
import json; d=json.load(open('gpurun_out/r05p_causal.json')); print('causal staged B=16', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'mem', round(d['peak_memory_GB'],1))"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_causal -o causal -- python $R/bench.py --config causal --steps 3 --warmup 1 > /tmp/prof_causal.log 2>&1
db=$(find /tmp/prof_causal -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db $R/gpurun_out/r05p_causal_kernel_stats.md 4
head -22 $R/gpurun_out/r05p_causal_kernel_stats.md | cut -c1-150
