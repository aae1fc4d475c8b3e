# r03r: f16x3 weight gradient fetching whole 128-byte lines (8 rows x 32 frames per DMA instruction); DPRNN-TasNet kernel trace
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "wgrad or golden or oracle" 2>&1 | tail -3 ) > gpurun_out/r03r_gputests.txt; cat gpurun_out/r03r_gputests.txt
{
echo "== timing"; python tools/gemm_bench.py --only W1,W2,W3,W4 --reps 30 2>&1 | grep "^W"
( cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/wg_line -- python $R/tools/gemm_bench.py --only W2,W3 --reps 3 > /tmp/wg.log 2>&1
  echo "== FETCH_SIZE (KiB / 2 per dispatch)"; python $R/tools/pmc_summary.py /tmp/wg_line "%pw_wgrad%" )
} > gpurun_out/r03r_wgrad_line.txt 2>&1; cat gpurun_out/r03r_wgrad_line.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step  wgrad", round(1e3*d["roofline_wgrad"]["avg_launch_ms"],1), "wg-heads", round(k["wgrad heads"]["avg_us"],1), "wg-conv1", round(k["wgrad conv1"]["avg_us"],1), "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" half-lines)
  python bench.py $B 2>gpurun_out/r03r_new.err | python -c "$summ" whole-lines
done
} > gpurun_out/r03r_ab.txt 2>&1; cat gpurun_out/r03r_ab.txt; tail -3 gpurun_out/r03r_new.err
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_dprnn -o dprnn -- python $R/bench.py --config dprnn --steps 6 --warmup 2 > /tmp/dprnn.log 2>&1
  f=$(find /tmp/prof_dprnn -name '*kernel_stats.csv' | head -1); cp $f $R/gpurun_out/r03r_dprnn_kernel_stats.csv; grep '^{' /tmp/dprnn.log | tail -1 | cut -c1-300 )
head -25 gpurun_out/r03r_dprnn_kernel_stats.csv | cut -c1-160
