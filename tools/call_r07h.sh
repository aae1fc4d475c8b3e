# r07h: full GPU test tier on the round-5 tree, float4 form of the gLN-sums kernel A/B, default bench (hipGraph replay), kernel trace
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
for v in 0 1 0 1; do SEPK_GLNW_F4=$v timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gln sums float4 $v: ms/step', round(d['ms_per_step'],3), d['config']['launch'])"; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r07h_gputests.txt; cat gpurun_out/r07h_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r07h_bench.out 2> gpurun_out/r07h_bench.err; echo rc $?; tail -c 300 gpurun_out/r07h_bench.err
tail -n 1 gpurun_out/r07h_bench.out > gpurun_out/r07h_bench.json; wc -c gpurun_out/r07h_bench.json; cut -c1-700 gpurun_out/r07h_bench.json
cp profiles/bench_detail.json gpurun_out/r07h_bench_detail.json
