# r03e: producer-side publish of the gLN-backward means, 1024-thread sums kernel, float4 slab stores of the weight gradient; FETCH_SIZE with time-separated half lines
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
{
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d /tmp/fc_$c -- /tmp/fetch_calib > /tmp/fc_$c.log 2>&1; tail -2 /tmp/fc_$c.log
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/fc_$c "%"
done
cd $GRAFT_REPO_ROOT
} > gpurun_out/r03e_fetch_calib.txt 2>&1; cat gpurun_out/r03e_fetch_calib.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03e_gputests.txt; cat gpurun_out/r03e_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" prev)
  python bench.py $B --no-pmc --no-stock 2>/dev/null | python -c "$summ" new
done
} > gpurun_out/r03e_ab.txt; cat gpurun_out/r03e_ab.txt
bash tools/profile_step.sh r03e 6 2>&1 | tail -2
head -36 gpurun_out/r03e_kernel_stats.md
