# r07zb: final tree -- smoke(), the headline bench line with all its legs (default flags), rocprofv3 kernel summary of the same command
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py > gpurun_out/r07zb_bench.log 2>gpurun_out/r07zb_bench.err ) 2>&1 | grep real
tail -n 1 gpurun_out/r07zb_bench.log > gpurun_out/r07zb_bench.json
cp profiles/bench_detail.json gpurun_out/r07zb_bench_detail.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r07zb_bench.json')); print('headline', round(d['ms_per_step'],3), 'ms', round(d['value']), d['config'].get('final_loss'), 'roofline', d['roofline']['kernel'][:40], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('roofline_family'), 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'], 'len', len(open('gpurun_out/r07zb_bench.json').read()))"
bash tools/profile_step.sh r07zb 8 2>&1 | tail -3
