# r05zc: the tree with the batched DMA issue: full GPU tests, smoke, the default bench as the driver runs it, kernel trace of the step,
# SepFormer's kernel trace (what its 124 ms are)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r05zc_gputests.txt; cat gpurun_out/r05zc_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r05zc_bench.out 2> gpurun_out/r05zc_bench.err; echo rc $?; tail -c 300 gpurun_out/r05zc_bench.err
tail -n 1 gpurun_out/r05zc_bench.out > gpurun_out/r05zc_bench.json; wc -c gpurun_out/r05zc_bench.json; cut -c1-700 gpurun_out/r05zc_bench.json
cp profiles/bench_detail.json gpurun_out/r05zc_bench_detail.json
bash tools/profile_step.sh r05zc 8 2>&1 | tail -3
head -24 gpurun_out/r05zc_kernel_stats.md | cut -c1-170
( cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_sf -o sf -- python $R/bench.py --config sepformer --steps 4 --warmup 2 > /tmp/sf.log 2>&1
  db=$(find /tmp/prof_sf -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r05zc_sepformer 6 )
head -26 gpurun_out/r05zc_sepformer_kernel_stats.md | cut -c1-170
