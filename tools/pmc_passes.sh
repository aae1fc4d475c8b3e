cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SEPK_SIDE_STREAM=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $pass -d /tmp/pmc_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock > /tmp/pmc_$tag.log 2>&1
  echo "=== $tag rc=$?"; tail -2 /tmp/pmc_$tag.log | cut -c1-200
  python $R/tools/pmc_summary.py /tmp/pmc_$tag "%" | grep -v "at::native\|rocclr\|elementwise" > $R/gpurun_out/pmc_$tag.txt
  wc -l $R/gpurun_out/pmc_$tag.txt
done
