R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for t in new old; do
  if [ $t = old ]; then export SEPK_LIN_TI=128 SEPK_LIN_TJ=128; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$t -o bench -- python $R/bench.py --config sepformer --steps 6 --warmup 2 > /tmp/prof_$t.log 2>&1
  grep '^{' /tmp/prof_$t.log | tail -1 | cut -c1-160
  db=$(find /tmp/prof_$t -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r07zg_sepformer_$t 8 > /dev/null
  echo "== $t"; grep -E "linear_kernel|attn|pw_gemm|GPU kernel time" $R/gpurun_out/r07zg_sepformer_${t}_kernel_stats.md | cut -c1-120
done
