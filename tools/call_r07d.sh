# r07d: depth of the cooperative kernel's raw X ring (one 4 KB chunk per stage and workgroup; depth 2 = what round 4 ran): are conv1 / heads^T /
# bottleneck^T waiting for their DMA?  (their workgroups live 20 - 33 us for 1.4 - 2.7 us of MFMAs)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
for ns in 2 3 4 6 8 2 6; do echo "== SEPK_COOP_NS=$ns"; SEPK_COOP_NS=$ns timeout 120 python tools/gemm_bench.py --packed --only F2,G3p,G3,G1,P1 --reps 20 2>&1 | grep "^[FGP]" | cut -c1-110; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -2
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
for v in 2 6 8 2 6 4; do SEPK_COOP_NS=$v SEPK_SIDE_STREAM=0 timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('coop ring $v, side off: ms/step', round(d['ms_per_step'],3))"; done
