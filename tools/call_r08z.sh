# r08z: the round's record on the final tree (one box): whole GPU tier, the default bench line with all its legs, a kernel trace, every --config
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -n 8 > gpurun_out/r08z_gputests.txt
tail -n 3 gpurun_out/r08z_gputests.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r08z_err.txt | tail -n 1 > gpurun_out/r08z_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r08z_bench.json')); print('headline', round(d['ms_per_step'],3), d['config']['launch'], d.get('fp32_mfma_pass'), d.get('step_roofline'), d.get('roofline_family'))"
cp profiles/bench_detail.json gpurun_out/r08z_bench_detail.json
bash tools/profile_step.sh r08z 10 2>&1 | tail -n 2
for c in causal dprnn dptnet galrnet sepformer; do
timeout 300 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r08z_bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r08z_bench_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['roofline']['bound'], round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
timeout 300 python bench.py --config sinkpit4 --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r08z_bench_sinkpit4.json; python -c "
import json; d=json.load(open('gpurun_out/r08z_bench_sinkpit4.json')); print('sinkpit4', round(d['ms_per_step'],2), 'ms', d['config']['launch'][:20], d.get('roofline',{}).get('frac'), d['config'].get('final_loss'))"
for b in fourier pinv; do timeout 300 python bench.py --basis $b --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r08z_bench_$b.json; python -c "
import json; d=json.load(open('gpurun_out/r08z_bench_$b.json')); print('$b', round(d['ms_per_step'],3), 'ms', d['config']['launch'][:20], d['config'].get('final_loss'))"; done
for b in 2 4; do timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>/dev/null | tail -n 1 > gpurun_out/r08z_bench_b$b.json; python -c "
import json; d=json.load(open('gpurun_out/r08z_bench_b$b.json')); print('batch $b', round(d['ms_per_step'],3), 'ms', d['config']['launch'][:20])"; done
