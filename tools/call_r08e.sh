# r08e: the recorded step under the exchange path (RCCL, one rank), the recipe trainer with auto_record, the whole GPU tier, headline bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 15 > gpurun_out/r08e_gputests.txt
tail -n 6 gpurun_out/r08e_gputests.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>gpurun_out/r08e_err.txt | tail -n 1 > gpurun_out/r08e_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r08e_bench.json')); print('headline', round(d['ms_per_step'],3), 'ms', d['config']['launch'], d['config'].get('final_loss'))"
