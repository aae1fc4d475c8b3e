# r07zk: key-major attention backward kernel in workgroups of 4 waves (3 waves per SIMD at D = 16) against 8
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for kv in 0 1 0 1; do echo "== SEPK_ATT_KV4=$kv"; SEPK_ATT_KV4=$kv timeout 200 python tools/attn_bench.py dptnet-intra sepformer-intra; done | tee gpurun_out/r07zk_attention_kv4.txt
SEPK_ATT_KV4=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | grep -E "passed|failed" | tail -2
for kv in 0 1; do SEPK_ATT_KV4=$kv timeout 300 python bench.py --config dptnet --steps 8 --warmup 3 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('kv4=$kv dptnet', round(d['ms_per_step'],2), 'ms', d['config'].get('final_loss'))"; done
