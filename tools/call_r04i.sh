# r04i: where the DPTNet and GALRNet steps spend their time (kernel traces)
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for cfg in dptnet galrnet; do
( cd /tmp && export TMPDIR=/tmp
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o $cfg -- python $R/bench.py --config $cfg --steps 4 --warmup 2 > /tmp/$cfg.log 2>&1
  db=$(find /tmp/prof_$cfg -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r04i_$cfg 6 )
head -16 gpurun_out/r04i_${cfg}_kernel_stats.md | cut -c1-140
done
