# r05zq: device fuzz of the round's new kernels (chained cLN far beyond the chip's resident workgroups, token gLN, attention)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 400 python tools/gpu_fuzz_round4.py 150 2>&1 | tail -25
