# r07a: first call of round 5 -- the tree as round 4 left it: batch sweep of the headline step (is a step of 16 utterances cheaper as
# two passes of 8?  the layer's working set at 8 utterances fits the 256 MB memory-side cache), side stream on / off.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock --steps 20 --warmup 5"
for b in 16 8 4 16 8 32; do
  timeout 200 python bench.py $Q --batch $b 2>/dev/null | tail -n 1 > gpurun_out/r07a_b$b.json
  python -c "
import json; d=json.load(open('gpurun_out/r07a_b$b.json')); print('batch', $b, 'ms/step', round(d['ms_per_step'],3), 'ms/utt', round(d['ms_per_step']/$b,4), 'frames/s', round(d['value']))"
done
SEPK_SIDE_STREAM=0 timeout 200 python bench.py $Q --batch 16 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('side stream off, batch 16: ms/step', round(d['ms_per_step'],3))"
SEPK_SIDE_STREAM=0 timeout 200 python bench.py $Q --batch 8 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('side stream off, batch 8: ms/step', round(d['ms_per_step'],3))"
