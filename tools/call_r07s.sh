# r07s: attention tiles for 64 / 128 steps + idle-wave exits: kernel tests, SepFormer / GALRNet / DPTNet benches, per-grid durations on SepFormer
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R/dnn-based_source_separation_amd/src
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "sibling or sepformer or dptnet" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for c in sepformer galrnet dptnet; do
timeout 300 python bench.py --config $c --steps 8 --warmup 3 2>/dev/null | tail -n 1 > gpurun_out/r07s_bench_$c.json; python -c "
import json; d=json.load(open('gpurun_out/r07s_bench_$c.json')); print('$c', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'roofline', d['roofline']['bound'], round(d['roofline']['achieved'],1), d['roofline']['unit'], round(d['roofline']['frac'],3), d['config'].get('final_loss'))"
done
cd /tmp && export TMPDIR=/tmp
for c in sepformer; do
  timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_$c -o bench -- python $R/bench.py --config $c --steps 4 --warmup 2 > /tmp/prof_$c.log 2>&1
  db=$(find /tmp/prof_$c -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $db $R/gpurun_out/r07s_${c}_attn.md 6 attn_ > /dev/null
  echo "== $c"; sed -n '/launches of kernels/,$p' $R/gpurun_out/r07s_${c}_attn.md
done
