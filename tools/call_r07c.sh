# r07c: the two depthwise kernels as persistent grids (consecutive rows per workgroup, next row in flight) against one workgroup per row
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
for v in "0 0" "1 1" "0 0" "1 1"; do set -- $v; echo "== persist fwd=$1 bwd=$2"; SEPK_DWF_PERSIST=$1 SEPK_DWB_PERSIST=$2 timeout 120 python tools/stream_bench.py 2>&1 | tail -9; done
echo "== bwd 2 workgroups per unit"; SEPK_DWB_WGS=2 timeout 120 python tools/stream_bench.py 2>&1 | tail -1
echo "== fwd 4 / 8 per unit"; SEPK_DWF_WGS=4 timeout 120 python tools/stream_bench.py 2>&1 | tail -1; SEPK_DWF_WGS=8 timeout 120 python tools/stream_bench.py 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "dwconv or depthwise" 2>&1 | tail -3
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
for v in "0 0 1" "1 1 1" "0 0 0" "1 1 0" "1 1 1" "0 0 1"; do set -- $v; SEPK_DWF_PERSIST=$1 SEPK_DWB_PERSIST=$2 SEPK_SIDE_STREAM=$3 timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persist $1 $2 side $3: ms/step', round(d['ms_per_step'],3))"; done
