# r05zm: final tree of the round (ABI 21, cLN 8x8 backward, token gLN slices, staged heads on f16x3): full GPU tests, smoke, default bench, causal / sibling benches, kernel trace of the causal step
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05zm_gputests.txt; cat gpurun_out/r05zm_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r05zm_bench.out 2> gpurun_out/r05zm_bench.err; echo rc $?; tail -c 300 gpurun_out/r05zm_bench.err
tail -n 1 gpurun_out/r05zm_bench.out > gpurun_out/r05zm_bench.json; wc -c gpurun_out/r05zm_bench.json; cut -c1-330 gpurun_out/r05zm_bench.json
cp profiles/bench_detail.json gpurun_out/r05zm_bench_detail.json
for c in causal dprnn dptnet galrnet sepformer; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zm_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r05zm_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'))"
done
( cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o c -- python $R/bench.py --config causal --steps 4 --warmup 2 > /tmp/c.log 2>&1
  db=$(find /tmp/prof_c -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r05zm_causal 6 )
head -22 gpurun_out/r05zm_causal_kernel_stats.md | cut -c1-150
