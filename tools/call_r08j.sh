# r08j: what heads^T costs with the row-sum epilogue (second read of z) against the plain product; heads weight gradient with the gLN prologue
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for rep in 1 2; do
python tools/gemm_bench.py --packed --reps 30 --only G3 2>&1 | tail -n 4 | tee -a gpurun_out/r08j_summary.txt
SEPK_COOP_MI4=0 python tools/gemm_bench.py --packed --reps 30 --only G3 2>&1 | tail -n 4 | tee -a gpurun_out/r08j_summary.txt
python tools/gemm_bench.py --packed --reps 30 --only W3 2>&1 | tail -n 3 | tee -a gpurun_out/r08j_summary.txt
done
