# r07e: the new GPU tests (layer kernels vs the oracle, RCCL on one rank, side stream on / off), the bench line with roofline_family, one --config line
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "tcn_layer" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "rccl or side_stream or graph_captured" 2>&1 | tail -8
timeout 600 python bench.py > gpurun_out/r07e_bench.out 2> gpurun_out/r07e_bench.err; echo rc $?; tail -c 400 gpurun_out/r07e_bench.err
tail -n 1 gpurun_out/r07e_bench.out > gpurun_out/r07e_bench.json; wc -c gpurun_out/r07e_bench.json; cat gpurun_out/r07e_bench.json
cp profiles/bench_detail.json gpurun_out/r07e_bench_detail.json
timeout 300 python bench.py --config causal --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07e_causal.json; cut -c1-1200 gpurun_out/r07e_causal.json
