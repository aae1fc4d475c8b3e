# r07r: attention kernels by grid (intra- vs inter-chunk shapes) on SepFormer / DPTNet / GALRNet
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in sepformer dptnet galrnet; do
  timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_$c -o bench -- python $R/bench.py --config $c --steps 4 --warmup 2 > /tmp/prof_$c.log 2>&1
  db=$(find /tmp/prof_$c -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $db $R/gpurun_out/r07r_${c}_attn.md 6 attn_ > /dev/null
  echo "== $c"; sed -n '/launches of kernels/,$p' $R/gpurun_out/r07r_${c}_attn.md
done
