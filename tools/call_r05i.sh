# r05i: direct at three workgroups per CU, `a` kept in registers, two-round-trip post
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
( timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -k "dwconv" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-900 | head -6 )
echo "== plain (sums)"; DWB_MODE=sums timeout 100 python tools/stream_bench.py 2>&1 | tail -2
for dbg in 0 7 3; do echo "== direct dbg=$dbg"; SEPK_DWB_DEBUG=$dbg DWB_MODE=direct timeout 100 python tools/stream_bench.py 2>&1 | tail -2; done
