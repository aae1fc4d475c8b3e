# r08k: soak of the recorded step: 300 steps x 3 runs at 4 utterances and 100 x 2 at 16, against the eager step on the same data
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 1200 python tools/seq_soak.py --steps 300 --batch 4 --runs 3 2>&1 | grep "^run" | tee gpurun_out/r08k_soak.txt
timeout 1200 python tools/seq_soak.py --steps 100 --batch 16 --runs 2 2>&1 | grep "^run" | tee -a gpurun_out/r08k_soak.txt
