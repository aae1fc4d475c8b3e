# r04y: last tree of the round: full GPU suite, smoke, default bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r04y_gputests.txt; cat gpurun_out/r04y_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04y_bench.json 2> gpurun_out/r04y_bench.err; tail -c 300 gpurun_out/r04y_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04y_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step', d['ms_per_step'], 'value', d['value'], 'roofline', r['bound'], r['frac'], 'avg ms', r['avg_launch_ms'])
print('wgrad', {k: d['roofline_wgrad'].get(k) for k in ('bound','frac','avg_launch_ms')}, 'traffic GB', (d.get('hbm_traffic') or {}).get('step_total_GB'))
k=d['roofline_by_kernel']; print({n: round(k[n]['avg_us'],1) for n in ('gemm heads','gemm conv1^T','gemm heads^T','gemm conv1','wgrad heads','wgrad conv1','depthwise bwd','depthwise fwd')})
PY
