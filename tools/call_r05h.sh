# r05h: where does the direct depthwise backward spend its time?  (SEPK_DWB_DEBUG: 1 no wait, 2 no post, 4 no P2)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
( timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -k "dwconv_bwd_direct_many" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-900 | head -6 )
echo "== plain (sums)"; DWB_MODE=sums timeout 100 python tools/stream_bench.py 2>&1 | tail -3
for dbg in 0 1 2 3 7; do echo "== direct dbg=$dbg"; SEPK_DWB_DEBUG=$dbg DWB_MODE=direct timeout 100 python tools/stream_bench.py 2>&1 | tail -3; done
