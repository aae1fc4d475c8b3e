# r05t: the tree as the driver will see it: full GPU suite, smoke, default bench
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r05t_gputests.txt; cat gpurun_out/r05t_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r05t_bench.out 2> gpurun_out/r05t_bench.err; echo rc $?; tail -c 300 gpurun_out/r05t_bench.err; tail -n 1 gpurun_out/r05t_bench.out | cut -c1-700
cp profiles/bench_detail.json gpurun_out/r05t_bench_detail.json
