# r08n: ceiling probe -- the weight-gradient kernels without their per-chunk workgroup barriers (WRONG results; timing only)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for rep in 1 2; do
python tools/gemm_bench.py --packed --reps 30 --only W 2>&1 | grep "^W[23]" | sed 's/^/tree   /' | tee -a gpurun_out/r08n_summary.txt
SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels_nobar.so python tools/gemm_bench.py --packed --reps 30 --only W 2>&1 | grep "^W[23]" | sed 's/^/nobar  /' | tee -a gpurun_out/r08n_summary.txt
done
