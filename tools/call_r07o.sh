# r07o: is the replay hazard the runtime's graph packet capture?  ten replayed runs with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
export SEPK_GRAPH=1
for i in 1 2 3 4 5 6 7 8; do DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('packet capture off, run $i', d['config']['launch'], round(d['ms_per_step'],3), d['config']['final_loss'])"; done
for i in 1 2 3 4; do AMD_SERIALIZE_KERNEL=3 timeout 200 python bench.py $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('serialize kernel, run $i', d['config']['launch'], round(d['ms_per_step'],3), d['config']['final_loss'])"; done
