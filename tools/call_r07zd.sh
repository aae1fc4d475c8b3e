# r07zd: dense kernel with two LDS tile pairs (one barrier per chunk) against one pair
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear or lstm" 2>&1 | grep -E "passed|failed" | tail -2
for nb in 1 2 1 2; do echo "== SEPK_LIN_NBUF=$nb"; SEPK_LIN_NBUF=$nb timeout 300 python tools/linear_bench.py; done | tee gpurun_out/r07zd_linear.txt
for nb in 1 2; do for c in dprnn sepformer; do
SEPK_LIN_NBUF=$nb timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07zd_${c}_$nb.json; python -c "
import json; d=json.load(open('gpurun_out/r07zd_${c}_$nb.json')); print('nbuf $nb $c', round(d['ms_per_step'],2), 'ms', d['config'].get('final_loss'))"
done; done
