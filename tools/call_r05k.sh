# r05k: (1) the B=16 fp64 reference fixture test; (2) experiment: the batch as independent sub-batches on several streams
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_model.py -x -q -s -k "batch16_against_the_fp64" 2>&1 | grep -E "batch-16|passed|failed|Error|assert" | cut -c1-300 | head ) | tee gpurun_out/r05k_b16.txt
timeout 300 python tools/experiments/two_streams.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05k_two_streams.txt
