# r05w: GALRNet's globally attentive block on token-major rows: golden parity, bench, kernel trace of one step
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "sibling or galrnet" 2>&1 | tail -2 )
for c in galrnet sepformer; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05w_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r05w_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config']['final_loss'])"
done
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r05w_prof -o galr -- python bench.py --config galrnet --steps 4 --warmup 2 > /dev/null 2>&1
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/r05w_prof/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:22]:
    print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs']) / 1e6, 2), round(float(r['AverageNs']) / 1e3, 1))
P
rm -rf gpurun_out/r05w_prof
