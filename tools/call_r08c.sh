# r08c: (1) the gLN2-sums kernel on a side stream beside heads^T (SEPK_SIDE_STREAM=2, eager launches) against one stream; (2) rocprofv3 kernel
# trace of the recorded step (profiles/r08c_kernel_stats.md) with the GPU-busy figure; (3) the sqnorm change
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for rep in 1 2; do for ss in 0 2; do
  SEPK_SIDE_STREAM=$ss timeout 300 python bench.py --eager --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>gpurun_out/r08c_err_$ss.txt | tail -n 1 > gpurun_out/r08c_bench_ss$ss.json
  python -c "
import json; d=json.load(open('gpurun_out/r08c_bench_ss$ss.json')); print('eager side-stream mode $ss', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08c_summary.txt
done; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1 > gpurun_out/r08c_bench_seq.json
python -c "
import json; d=json.load(open('gpurun_out/r08c_bench_seq.json')); print('recorded sequence', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08c_summary.txt
bash tools/profile_step.sh r08c 10 2>&1 | tail -n 3
