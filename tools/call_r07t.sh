cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 300 python tools/attn_bench.py
