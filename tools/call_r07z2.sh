cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 300 python tools/find_copies.py 2>&1 | tail -45 | tee gpurun_out/r07z2_copies.txt
