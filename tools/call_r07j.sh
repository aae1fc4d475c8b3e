# r07j: the other workloads' bench lines (each with its roofline); causal as graph replay vs eager; sinkpit4
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for g in 1 0; do SEPK_GRAPH=$g timeout 300 python bench.py --config causal --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07j_causal_g$g.json; python -c "
import json; d=json.load(open('gpurun_out/r07j_causal_g$g.json')); print('causal graph=$g', round(d['ms_per_step'],2), 'ms', d['config']['launch'][:60], 'frac', round(d['roofline']['frac'],3), d['config'].get('final_loss'))"; done
for c in dprnn dptnet galrnet sepformer; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07j_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r07j_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'roofline', d['roofline']['bound'], round(d['roofline']['achieved'],1), d['roofline']['unit'], round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
timeout 300 python bench.py --config sinkpit4 --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r07j_sinkpit4.json; python -c "
import json; d=json.load(open('gpurun_out/r07j_sinkpit4.json')); print('sinkpit4', round(d['ms_per_step'],2), 'ms', d['config']['launch'], d.get('roofline',{}).get('frac'))"
