# r07zn: cLN chain forward with 64-frame tiles (256-byte row pieces) against 32
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for tw in 32 64 32 64; do echo -n "TW=$tw  "; SEPK_CLN_TW=$tw timeout 200 python tools/cln_bench.py; done | tee gpurun_out/r07zn_cln_tw.txt
SEPK_CLN_TW=64 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "cln" 2>&1 | grep -E "passed|failed" | tail -2
for tw in 32 64; do SEPK_CLN_TW=$tw timeout 300 python bench.py --config causal --steps 8 --warmup 3 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TW=$tw causal', round(d['ms_per_step'],2), 'ms', d['config'].get('final_loss'))"; done
