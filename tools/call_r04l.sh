# r04l: full GPU suite + smoke on the last tree
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/r04l_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r04l_gputests.txt
