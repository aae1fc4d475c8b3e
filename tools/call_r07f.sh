cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "tcn_layer" 2>&1 | grep -E "AssertionError|max err|passed|failed" | head -20
