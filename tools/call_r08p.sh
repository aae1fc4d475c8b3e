# r08p: last check of the final tree: GPU tier, smoke, default bench line
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -n 4 | tee gpurun_out/r08p_summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 | tee -a gpurun_out/r08p_summary.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/r08p_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r08p_bench.json')); print('headline', round(d['ms_per_step'],3), d['config']['launch'], d['fp32_mfma_pass'], d['roofline']['frac'], d['roofline']['frac_8d'])" | tee -a gpurun_out/r08p_summary.txt
