# r07z: the round's measurement recipe on the final tree: GPU tier, smoke, default bench (all legs), rocprofv3 kernel trace of the same command
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r07z_gputests.txt; cat gpurun_out/r07z_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r07z_bench.out 2> gpurun_out/r07z_bench.err; echo rc $?; tail -c 300 gpurun_out/r07z_bench.err
tail -n 1 gpurun_out/r07z_bench.out > gpurun_out/r07z_bench.json; wc -c gpurun_out/r07z_bench.json; cut -c1-400 gpurun_out/r07z_bench.json
cp profiles/bench_detail.json gpurun_out/r07z_bench_detail.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_z -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --no-graph > /tmp/prof_z.log 2>&1
echo "trace rc=$?"; grep '^{' /tmp/prof_z.log | tail -1 | cut -c1-200
db=$(find /tmp/prof_z -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db $R/gpurun_out/r07z_kernel_stats.md 10
