cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 400 python tools/gpu_fuzz_round5.py 150 2>&1 | tail -15 | tee gpurun_out/r07y_fuzz.txt
