cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
SEPK_GRAPH=1 timeout 300 python bench.py --config causal --steps 4 --warmup 2 2>&1 | tail -25 | cut -c1-300
