# The measurement recipe behind profiles/r02*: full GPU test suite, bench JSON, rocprofv3 kernel trace, the PMC passes and the traffic
# table, all written to gpurun_out/<tag>_*.  Run on the GPU box:  gpurun -- 'bash tools/measure_all.sh r02x'
cd $GRAFT_REPO_ROOT
tag=${1:-r02g}
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${tag}_gputests.txt; cat gpurun_out/${tag}_gputests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 300 gpurun_out/${tag}_bench.err; python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); r=d['roofline']
print('ms/step', d['ms_per_step'], 'roofline', r['bound'], r['achieved'], r['peak'], r['frac'], 'avg ms', r['avg_launch_ms'], 'f32', d['fp32_mfma_pass']['ms_per_step'], d['fp32_mfma_pass']['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('batch16'), 'wgrad', d.get('roofline_wgrad',{}).get('frac'))"
bash tools/profile_step.sh $tag 8 2>&1 | tail -3
bash tools/pmc_passes.sh 2>&1 | tail -6
python tools/pmc_traffic.py gpurun_out gpurun_out/$tag 3
