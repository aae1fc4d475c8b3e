mkdir -p $GRAFT_REPO_ROOT/gpurun_out
bash $GRAFT_REPO_ROOT/tools/kernel_power.sh 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r07zi_kernel_power.txt
