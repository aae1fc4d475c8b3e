# r08r: ceiling probe -- half of every consumer wave's G rows as if pre-split (what pre-splitting dS once per step could give the heads' weight gradient)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
P=$PWD/dnn-based_source_separation_amd
for rep in 1 2 3; do
python tools/gemm_bench.py --packed --reps 30 --only W3 2>&1 | grep "^W3" | sed 's/^/tree       /' | tee -a gpurun_out/r08r_summary.txt
SEPKERNELS_LIB=$P/libsepkernels_halfsplit.so python tools/gemm_bench.py --packed --reps 30 --only W3 2>&1 | grep "^W3" | sed 's/^/half split /' | tee -a gpurun_out/r08r_summary.txt
done
