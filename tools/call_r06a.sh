# r06a: last tree of round 4 (after the SepFormer feed-forward move): full GPU tests, smoke, default bench, the other workloads
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06a_gputests.txt; cat gpurun_out/r06a_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r06a_bench.out 2> gpurun_out/r06a_bench.err; echo rc $?; tail -c 300 gpurun_out/r06a_bench.err
tail -n 1 gpurun_out/r06a_bench.out > gpurun_out/r06a_bench.json; wc -c gpurun_out/r06a_bench.json; cut -c1-330 gpurun_out/r06a_bench.json
cp profiles/bench_detail.json gpurun_out/r06a_bench_detail.json
for c in causal dprnn dptnet galrnet sepformer; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r06a_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r06a_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
done
