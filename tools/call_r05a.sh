# r05a: round-4 baseline on this round's first box: default bench (compact line + detail), as the driver runs it
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
python bench.py > gpurun_out/r05a_bench.out 2> gpurun_out/r05a_bench.err; echo rc $?; tail -c 400 gpurun_out/r05a_bench.err
tail -n 1 gpurun_out/r05a_bench.out | wc -c
tail -n 1 gpurun_out/r05a_bench.out
cp profiles/bench_detail.json gpurun_out/r05a_bench_detail.json
