# r07g: hipGraph replay of the one-stream step against eager launches (side stream off = the new default)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
for g in 0 1 0 1; do SEPK_GRAPH=$g timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph $g: ms/step', round(d['ms_per_step'],3), d['config']['launch'])"; done
for b in 8 4; do for g in 0 1; do SEPK_GRAPH=$g timeout 200 python bench.py $Q --batch $b 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch $b graph $g: ms/step', round(d['ms_per_step'],3))"; done; done
