# r04z (final tree of the round): the round's measurement: full GPU suite, default bench (PMC traffic, CPU reference, stock-torch and inference legs), kernel trace, the other configurations
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
tag=r04z
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/${tag}_gputests.txt; cat gpurun_out/${tag}_gputests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 400 gpurun_out/${tag}_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04z_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step', d['ms_per_step'], 'value', d['value'], 'roofline', r['bound'], r['achieved'], r['frac'], 'avg ms', r['avg_launch_ms'], 'traffic/launch', r.get('traffic'))
print('wgrad', {k: d['roofline_wgrad'].get(k) for k in ('bound','frac','avg_launch_ms','traffic')})
print('f32', d['fp32_mfma_pass']['ms_per_step'], 'cpu', d['cpu_baseline'], 'stock', {k: d.get('hipified_baseline',{}).get(k) for k in ('ms_per_step','value')})
h=d.get('hbm_traffic') or {}
print('traffic GB', h.get('step_total_GB'), h.get('over_algorithmic'), 'inference', d.get('inference'))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_step.sh $tag 8 2>&1 | tail -3
summ2='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), "ms/step", round(d["value"]), d["unit"], "loss", d["config"]["final_loss"], (d.get("roofline") or {}).get("frac"))'
for cfg in dprnn sinkpit4 galrnet dptnet sepformer; do python bench.py --config $cfg --steps 8 --warmup 3 2>gpurun_out/${tag}_$cfg.err > gpurun_out/${tag}_bench_$cfg.json; python -c "$summ2" $cfg < gpurun_out/${tag}_bench_$cfg.json; done 2>&1 | tee gpurun_out/${tag}_configs.txt
