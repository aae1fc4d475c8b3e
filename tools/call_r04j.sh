# r04j: allocator configuration (page mapping of the 7 GB of saved activations): does the step care?
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass --no-pmc --no-stock --no-kernel-timing"
summ='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],3), "ms/step loss", d["config"]["final_loss"])'
{
python bench.py $B 2>/dev/null | python -c "$summ" default
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python bench.py $B 2>/dev/null | python -c "$summ" expandable_segments || echo "expandable_segments: failed"
PYTORCH_NO_HIP_MEMORY_CACHING=0 PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:4096 python bench.py $B 2>/dev/null | python -c "$summ" max_split_4096 || echo failed
python bench.py $B 2>/dev/null | python -c "$summ" default
} 2>&1 | tee gpurun_out/r04j_alloc.txt
