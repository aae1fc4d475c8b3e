# r04c: the bench's multi-rank path (barriers, max over ranks, bucket timing, ranks block) with two ranks on ONE GPU over gloo
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
SEPK_BENCH_BACKEND=gloo SEPK_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r04c_two_ranks.json 2> gpurun_out/r04c_two_ranks.err
echo rc=$?; tail -c 600 gpurun_out/r04c_two_ranks.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04c_two_ranks.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print(json.dumps(d['ranks'])[:700])
PY
