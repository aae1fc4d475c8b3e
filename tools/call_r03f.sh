# r03f: two-level arrival counters; weight-gradient fetch traffic vs the row stride (ldt = 4096: rows a power of two apart; 4224: not)
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
cd /tmp && export TMPDIR=/tmp
for ldt in 4096 4224; do
  for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $c -d /tmp/wg_${ldt}_$tag -- python $R/tools/gemm_bench.py --only W2,W3 --reps 3 --ldt $ldt > /tmp/wg.log 2>&1
    echo "== ldt $ldt $tag"; grep "^W" /tmp/wg.log
    python $R/tools/pmc_summary.py /tmp/wg_${ldt}_$tag "%pw_wgrad%"
  done
done
cd $R
} > gpurun_out/r03f_wgrad_fetch.txt 2>&1; cat gpurun_out/r03f_wgrad_fetch.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03f_gputests.txt; cat gpurun_out/r03f_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], "ms/step  gemm", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], " wgrad", d["roofline_wgrad"]["avg_launch_ms"], d["roofline_wgrad"]["frac"], "loss", d["config"]["final_loss"])'
{
for rep in 1 2; do
  (cd _ab_prev && PYTHONPATH=dnn-based_source_separation_amd/src python bench.py $B 2>/dev/null | python -c "$summ" prev)
  python bench.py $B --no-pmc --no-stock 2>gpurun_out/r03f_new.err | python -c "$summ" new
done
} > gpurun_out/r03f_ab.txt 2>&1; cat gpurun_out/r03f_ab.txt; tail -3 gpurun_out/r03f_new.err
bash tools/profile_step.sh r03f 6 2>&1 | tail -2
head -32 gpurun_out/r03f_kernel_stats.md
