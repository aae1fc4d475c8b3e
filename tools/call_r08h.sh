# r08h: SinkPIT in the recorded step: GPU tests, sinkpit4 recorded against eager (one box)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu -k "record or sinkpit or criterion_kernels or absmax" 2>&1 | tail -n 4 | tee gpurun_out/r08h_summary.txt
for rep in 1 2; do for mode in seq eager; do
  extra=""; [ $mode = eager ] && extra="--eager"
  timeout 300 python bench.py --config sinkpit4 --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing $extra 2>/dev/null | tail -n 1 > gpurun_out/r08h_sinkpit4_$mode.json
  python -c "
import json; d=json.load(open('gpurun_out/r08h_sinkpit4_$mode.json')); print('sinkpit4 $mode', round(d['ms_per_step'],3), 'ms', d['config']['launch'][:24], d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08h_summary.txt
done; done
for b in 4; do for mode in seq eager; do
  extra=""; [ $mode = eager ] && extra="--eager"
  timeout 300 python bench.py --config sinkpit4 --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing $extra 2>/dev/null | tail -n 1 > gpurun_out/r08h_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/r08h_tmp.json')); print('sinkpit4 B=$b $mode', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08h_summary.txt
done; done
