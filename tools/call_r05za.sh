# r05za: staged weight-gradient kernel with the batched DMA issue (one asm statement per operand and pair) against the tree before it
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
export SEPK_WGRAD_RF=0
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | tail -3 )
for v in "" _head "" _head; do echo "== libsepkernels$v"; SEPKERNELS_LIB=$PWD/dnn-based_source_separation_amd/libsepkernels$v.so timeout 300 python tools/gemm_bench.py --only W --reps 20 2>&1 | grep "^W" | cut -c1-110; done
