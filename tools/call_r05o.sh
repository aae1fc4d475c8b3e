# r05o: kernel trace of the staged causal step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_causal -o causal -- python $R/bench.py --config causal --steps 3 --warmup 1 > /tmp/prof_causal.log 2>&1
echo rc=$?; tail -n 1 /tmp/prof_causal.log | cut -c1-300
db=$(find /tmp/prof_causal -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db $R/gpurun_out/r05o_causal_kernel_stats.md 4
head -30 $R/gpurun_out/r05o_causal_kernel_stats.md | cut -c1-190
