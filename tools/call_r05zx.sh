# r05zx: causal bench with the fused clip + Adam of the headline step against torch's foreach optimizer
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for ta in 0 1 0; do SEPK_TORCH_ADAM=$ta timeout 300 python bench.py --config causal --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r05zx_causal_$ta.json; python -c "
import json; d=json.load(open('gpurun_out/r05zx_causal_$ta.json')); print('causal torch_adam=$ta', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d['config'].get('final_loss'), d['config'].get('launch'))"; done
