# r07m: stability of the replayed step: ten runs of the default launch mode, final loss of each (must be the eager run's 0.08124 every time)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('run $i', d['config']['launch'], round(d['ms_per_step'],3), d['config']['final_loss'])"; done
SEPK_GRAPH=0 timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('eager', d['config']['launch'], round(d['ms_per_step'],3), d['config']['final_loss'])"
timeout 200 python bench.py $Q --steps 10 --warmup 3 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('10+3', d['config']['launch'], round(d['ms_per_step'],3), d['config']['final_loss'])"
