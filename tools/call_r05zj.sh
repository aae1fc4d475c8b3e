# r05zj: kernel traces of the dual-path steps after the token-major work (what is left in DPTNet / GALRNet / DPRNN-TasNet)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in dptnet galrnet dprnn; do
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o $c -- python $R/bench.py --config $c --steps 4 --warmup 2 > /tmp/$c.log 2>&1
  db=$(find /tmp/prof_$c -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r05zj_$c 6 | tail -1
  head -20 $R/gpurun_out/r05zj_${c}_kernel_stats.md | cut -c1-140
done
