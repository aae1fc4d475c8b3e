# r03b: (1) GPU tests of the slotted gLN-backward sums, (2) step A/B against the previous commit on the same box, (3) FETCH_SIZE / WRITE_SIZE
# calibration on known byte counts, (4) weight-gradient kernel: raw-ring depth 2 / 3 / 4, timings and s_memtime stamps
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03b_gputests.txt; cat gpurun_out/r03b_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" prev)
  python bench.py $B 2>/dev/null | python -c "$summ" new
done
} > gpurun_out/r03b_ab.txt; cat gpurun_out/r03b_ab.txt
{
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d /tmp/fc_$c -- /tmp/fetch_calib > /tmp/fc_$c.log 2>&1; tail -2 /tmp/fc_$c.log
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/fc_$c "%"
done
cd $GRAFT_REPO_ROOT
} > gpurun_out/r03b_fetch_calib.txt 2>&1; cat gpurun_out/r03b_fetch_calib.txt
{
for ns in 2 3 4; do
  echo "== SEPK_WPC_NS=$ns"
  SEPK_WPC_NS=$ns python tools/gemm_bench.py --only W2,W3,W4,W1 --reps 30
  SEPK_WPC_NS=$ns SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels_wpcprof.so python tools/wpc_prof.py
done
} > gpurun_out/r03b_wgrad_ns.txt 2>&1; cat gpurun_out/r03b_wgrad_ns.txt
