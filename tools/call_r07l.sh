# r07l: the persistent producer / consumer kernel on the prologue-free short contractions (conv1, heads^T, bottleneck^T) against the cooperative kernel
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or tcn_layer" 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in 0 1 0 1; do echo "== SEPK_PC_PERSIST=$v"; SEPK_PC_PERSIST=$v timeout 120 python tools/gemm_bench.py --packed --only F2,G3p,G1,P1 --reps 20 2>&1 | grep "^[FGP]" | cut -c1-110; done
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
for v in 0 1 0 1; do SEPK_PC_PERSIST=$v timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persistent pc $v: ms/step', round(d['ms_per_step'],3), d['config']['launch'], d['config']['final_loss'])"; done
