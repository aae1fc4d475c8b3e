# r07zj: the full GPU tier on the final tree
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
