# r07n: the final tree's default bench (eager, one stream) with all legs, and its kernel trace
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r07n_bench.out 2> gpurun_out/r07n_bench.err; echo rc $?; tail -c 300 gpurun_out/r07n_bench.err
tail -n 1 gpurun_out/r07n_bench.out > gpurun_out/r07n_bench.json; wc -c gpurun_out/r07n_bench.json; cut -c1-300 gpurun_out/r07n_bench.json
cp profiles/bench_detail.json gpurun_out/r07n_bench_detail.json
timeout 300 python bench.py > gpurun_out/r07n_bench_default.out 2>/dev/null; tail -n 1 gpurun_out/r07n_bench_default.out | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default flags:', d['ms_per_step'], d['config']['launch'], d['config']['final_loss'], d['steps'], d['warmup'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock > /tmp/prof_n.log 2>&1
echo "trace rc=$?"; grep '^{' /tmp/prof_n.log | tail -1 | cut -c1-200
db=$(find /tmp/prof_n -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db $R/gpurun_out/r07n_kernel_stats.md 10
ls /tmp/prof_n/*/ 2>/dev/null | head; f=$(find /tmp/prof_n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -30 $f > $R/gpurun_out/r07n_rocprof_kernel_stats.csv
