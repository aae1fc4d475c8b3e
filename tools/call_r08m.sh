# r08m: two ranks on ONE GPU over gloo (SEPK_BENCH_ONE_GPU=1): the recorded step in segments between the gradient buckets against the eager
# data-parallel step -- the N > 1 code path of bench.py on the hardware a test box has (not a scaling measurement)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for mode in seq eager; do
  extra=""; [ $mode = eager ] && extra="--eager"
  SEPK_BENCH_BACKEND=gloo SEPK_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --batch 8 --steps 10 --warmup 3 --no-kernel-timing $extra 2>gpurun_out/r08m_err_$mode.txt | tail -n 1 > gpurun_out/r08m_two_ranks_$mode.json
  python -c "
import json; d=json.load(open('gpurun_out/r08m_two_ranks_$mode.json')); print('2 ranks on one GPU, gloo, $mode:', round(d['ms_per_step'],2), 'ms', d['config'])" 2>&1 | tee -a gpurun_out/r08m_summary.txt
done
tail -n 3 gpurun_out/r08m_err_seq.txt
