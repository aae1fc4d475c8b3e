# r07zl: every --config of bench.py on the final tree (one box)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for c in causal dprnn dptnet galrnet sepformer; do
timeout 300 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07zl_bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r07zl_bench_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['roofline']['bound'], round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
timeout 300 python bench.py --config sinkpit4 --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r07zl_bench_sinkpit4.json; python -c "
import json; d=json.load(open('gpurun_out/r07zl_bench_sinkpit4.json')); print('sinkpit4', round(d['ms_per_step'],2), 'ms', d['config']['launch'], d.get('roofline',{}).get('frac'), d['config'].get('final_loss'))"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r07zl_bench_headline.json; python -c "
import json; d=json.load(open('gpurun_out/r07zl_bench_headline.json')); print('headline', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))"
for b in fourier pinv; do timeout 300 python bench.py --basis $b --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r07zl_bench_$b.json; python -c "
import json; d=json.load(open('gpurun_out/r07zl_bench_$b.json')); print('$b', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))"; done
