# r08o: ceiling probes -- (w) the weight-gradient consumers without the G split, (g) the producer / consumer GEMM's producers without the X split
# (operands as if they arrived pre-split from HBM: WRONG results, timing only)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
P=$PWD/dnn-based_source_separation_amd
for rep in 1 2; do
python tools/gemm_bench.py --packed --reps 30 --only W 2>&1 | grep "^W[23]" | sed 's/^/tree       /' | tee -a gpurun_out/r08o_summary.txt
SEPKERNELS_LIB=$P/libsepkernels_nosplitw.so python tools/gemm_bench.py --packed --reps 30 --only W 2>&1 | grep "^W[23]" | sed 's/^/no G split /' | tee -a gpurun_out/r08o_summary.txt
python tools/gemm_bench.py --packed --reps 30 --only "F3" 2>&1 | grep "^F3" | sed 's/^/tree       /' | tee -a gpurun_out/r08o_summary.txt
SEPKERNELS_LIB=$P/libsepkernels_nosplitg.so python tools/gemm_bench.py --packed --reps 30 --only "F3" 2>&1 | grep "^F3" | sed 's/^/no X split /' | tee -a gpurun_out/r08o_summary.txt
python tools/gemm_bench.py --packed --reps 30 --only "G2" 2>&1 | grep "^G2" | sed 's/^/tree       /' | tee -a gpurun_out/r08o_summary.txt
SEPKERNELS_LIB=$P/libsepkernels_nosplitg.so python tools/gemm_bench.py --packed --reps 30 --only "G2" 2>&1 | grep "^G2" | sed 's/^/no X split /' | tee -a gpurun_out/r08o_summary.txt
done
