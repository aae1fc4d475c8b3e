# r03s: per-chunk timeline of the whole-line f16x3 weight gradient (stamps); DPRNN-TasNet kernel trace
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
SEPKERNELS_LIB=$R/dnn-based_source_separation_amd/libsepkernels_wpcprof.so python tools/wpc16_prof.py 2>&1 | grep -v amdgpu > gpurun_out/r03s_wpc16_stamps.txt; cat gpurun_out/r03s_wpc16_stamps.txt
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_dprnn -o dprnn -- python $R/bench.py --config dprnn --steps 6 --warmup 2 > /tmp/dprnn.log 2>&1
  grep '^{' /tmp/dprnn.log | tail -1 | cut -c1-200
  find /tmp/prof_dprnn -type f | head
  db=$(find /tmp/prof_dprnn -name '*.db' | head -1)
  python $R/tools/export_profile.py $db $R/gpurun_out/r03s_dprnn 8 )
head -40 gpurun_out/r03s_dprnn_kernel_stats.md
