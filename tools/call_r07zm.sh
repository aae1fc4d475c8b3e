# r07zm: the multi-rank code path of bench.py on one GPU (two ranks share device 0, gloo carries the exchange): launcher, buckets, max over ranks
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
SEPK_BENCH_BACKEND=gloo SEPK_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --batch 8 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>gpurun_out/r07zm.err | tail -n 1 > gpurun_out/r07zm_bench_2ranks.json
python -c "
import json; d=json.load(open('gpurun_out/r07zm_bench_2ranks.json')); print('2 ranks on one GPU:', d['n_gpus'], round(d['ms_per_step'],2), 'ms', round(d['value']), d['config'].get('final_loss'), d['config'].get('parallelism'), {k: d[k] for k in d if 'comm' in k or 'rank' in k})"
tail -3 gpurun_out/r07zm.err
