# r03d: sums kernel fixed (loads batched), side-stream hand-off, one-thread means in the depthwise backward; new parity tests; the bench with its new legs
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03d_gputests.txt; cat gpurun_out/r03d_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" prev)
  python bench.py $B --no-pmc --no-stock 2>/dev/null | python -c "$summ" new
done
} > gpurun_out/r03d_ab.txt; cat gpurun_out/r03d_ab.txt
bash tools/profile_step.sh r03d 6 2>&1 | tail -2
head -34 gpurun_out/r03d_kernel_stats.md
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r03d_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03d_bench.json'))
print('ms/step', d['ms_per_step'], 'value', d['value'])
r=d['roofline']; print('roofline', r['frac'], r['avg_launch_ms'], r.get('traffic'), r.get('traffic_live'), r.get('traffic_over_algorithmic'))
print('wgrad', d['roofline_wgrad']['frac'], d['roofline_wgrad'].get('traffic_over_algorithmic'))
print('hbm_traffic', {k: v for k, v in d.get('hbm_traffic', {}).items() if k != 'per_kernel'})
print('stock', d.get('hipified_baseline')); print('cpu', d.get('cpu_baseline'))
for k, v in d['roofline_by_kernel'].items():
    print('  {:24s} n={:5.1f} {:7.1f} us  {:6.1f} MB  hbm {:.2f}  8d {:.2f}  {}'.format(k, v['launches_per_step'], v['avg_us'], v.get('algorithmic_MB_per_launch', 0), v.get('hbm_frac', 0), v.get('hbm_frac_8d', 0), v['bound']))
PY
