# r07q: sep_rownorm_* / sep_relu_drop_* on the device: kernel tests, the sibling models' fixtures, SepFormer / GALRNet benches and SepFormer's kernel table
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "rownorm or relu_drop or gln_tokens" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "sibling or sepformer" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for c in sepformer galrnet; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07q_bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r07q_bench_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'roofline', d['roofline']['bound'], round(d['roofline']['achieved'],1), d['roofline']['unit'], round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_sf -o bench -- python $R/bench.py --config sepformer --steps 6 --warmup 2 > /tmp/prof_sf.log 2>&1
db=$(find /tmp/prof_sf -name '*.db' | head -1)
python $R/tools/export_profile.py $db $R/gpurun_out/r07q_sepformer 8
