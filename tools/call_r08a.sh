# r08a: first GPU call of round 6 -- the recorded launch sequence (ABI 23): GPU tests of it, then eager vs recorded at 4 / 8 / 16 utterances on one box
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "recorded or record_refuses" 2>&1 | tail -n 15 > gpurun_out/r08a_tests_record.txt
cat gpurun_out/r08a_tests_record.txt
for b in 16 4 8; do for mode in seq eager; do
  extra=""; [ $mode = eager ] && extra="--eager"
  timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing $extra 2>gpurun_out/r08a_err_${b}_$mode.txt | tail -n 1 > gpurun_out/r08a_bench_${b}_$mode.json
  python -c "
import json; d=json.load(open('gpurun_out/r08a_bench_${b}_$mode.json')); print('B=$b $mode', round(d['ms_per_step'],3), 'ms', d['config']['launch'][:30], d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08a_summary.txt
done; done
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 15 > gpurun_out/r08a_gputests.txt
tail -n 5 gpurun_out/r08a_gputests.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r08a_err_full.txt | tail -n 1 > gpurun_out/r08a_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r08a_bench.json')); print('full', round(d['ms_per_step'],3), d['config']['launch'], d['roofline'], d.get('fp32_mfma_pass'), d.get('step_roofline'))"
cp profiles/bench_detail.json gpurun_out/r08a_bench_detail.json 2>/dev/null
