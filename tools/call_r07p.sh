# r07p: current per-kernel tables of the other workloads (which torch kernels are left on them)
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in sepformer dptnet galrnet dprnn causal; do
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o bench -- python $R/bench.py --config $c --steps 6 --warmup 2 > /tmp/prof_$c.log 2>&1
  echo "$c rc=$?"; grep '^{' /tmp/prof_$c.log | tail -1 | cut -c1-200
  db=$(find /tmp/prof_$c -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r07p_$c 8
done
