# r05u: `python bench.py --gpus 2` as the driver would call it, on a one-GPU box: both ranks on device 0, all-reduce through gloo
# (exercises launch_ranks, the rendezvous, the bucketed exchange, max-over-ranks timing and the compact line); then the GPU recipe tests
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
SEPK_BENCH_ONE_GPU=1 SEPK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r05u_two_ranks.out 2> gpurun_out/r05u_two_ranks.err; echo rc $?; tail -c 400 gpurun_out/r05u_two_ranks.err; tail -n 1 gpurun_out/r05u_two_ranks.out | cut -c1-900
python -c "
import json; d=json.load(open('profiles/bench_detail_n2.json')); print(d['ranks'])"
( timeout 600 python -m pytest tests/test_gpu_recipe.py -x -q 2>&1 | tail -2 )
