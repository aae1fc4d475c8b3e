# r03q: weight-gradient operand fetch: 16 rows x 64 B per DMA instruction (chunk pairs) vs 8 rows x 128 B (whole lines; probe build, results garbage)
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
LINE=$R/dnn-based_source_separation_amd/libsepkernels_line.so
{
for v in half line; do
  [ $v = line ] && export SEPKERNELS_LIB=$LINE
  echo "== $v: timing"; python tools/gemm_bench.py --only W1,W2,W3,W4 --reps 30 2>&1 | grep "^W"
  ( cd /tmp && export TMPDIR=/tmp
    timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/wg_$v -- python $R/tools/gemm_bench.py --only W2,W3 --reps 3 > /tmp/wg.log 2>&1
    echo "== $v: FETCH_SIZE (KiB / 2 per dispatch)"; python $R/tools/pmc_summary.py /tmp/wg_$v "%pw_wgrad%" )
done
unset SEPKERNELS_LIB
} > gpurun_out/r03q_wgrad_line.txt 2>&1; cat gpurun_out/r03q_wgrad_line.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline_by_kernel"]; print(sys.argv[1], round(d["ms_per_step"],3), "ms/step  wgrad", round(1e3*d["roofline_wgrad"]["avg_launch_ms"],1), "wg-heads", round(k["wgrad heads"]["avg_us"],1), "wg-conv1", round(k["wgrad conv1"]["avg_us"],1))'
{
for rep in 1 2; do
  python bench.py $B 2>/dev/null | python -c "$summ" half-lines
  SEPKERNELS_LIB=$LINE python bench.py $B 2>gpurun_out/r03q_new.err | python -c "$summ" whole-lines-probe
done
} > gpurun_out/r03q_ab.txt 2>&1; cat gpurun_out/r03q_ab.txt; tail -3 gpurun_out/r03q_new.err
