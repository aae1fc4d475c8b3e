# r07w: full GPU tier on the tree with sep_rownorm_* / sep_relu_drop_* / the one-pass attention forward / the raw stream accessor; headline bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/r07w_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r07w_bench.json')); print('headline', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'), d['roofline']['frac'], d.get('roofline_family'))"
for c in galrnet dptnet dprnn; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07w_bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r07w_bench_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
