cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for rep in 1 2 3; do python tools/gemm_bench.py --packed --reps 30 --only W3 2>&1 | grep "^W3" | tee -a gpurun_out/r08t_summary.txt; done
