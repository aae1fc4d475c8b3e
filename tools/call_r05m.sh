# r05m: SURVEY 8d config coverage: DPRNN-TasNet (configs[3]) at the recipe batch and at larger batches; Conv-TasNet 4-spk SinkPIT (configs[4]) at 200 and 10 iterations
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
for b in 2 8 16 32; do timeout 300 python bench.py --config dprnn --batch $b --steps 6 --warmup 2 2>/dev/null | tail -n 1 > gpurun_out/r05m_dprnn_b$b.json; python -c "
import json; d=json.load(open('gpurun_out/r05m_dprnn_b$b.json')); print('dprnn B=$b', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', 'frac', round(d['roofline']['frac'],3), 'mem', round(d['peak_memory_GB'],1))"; done
for k in 200 10; do timeout 300 python bench.py --config sinkpit4 --sink-iters $k --steps 10 --warmup 3 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>/dev/null | tail -n 1 > gpurun_out/r05m_sinkpit4_k$k.json; cp profiles/bench_detail.json gpurun_out/r05m_sinkpit4_k${k}_detail.json; python -c "
import json; d=json.load(open('gpurun_out/r05m_sinkpit4_k$k.json')); print('sinkpit4 k=$k', round(d['ms_per_step'],2), 'ms', round(d['value']), 'frames/s', d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))"; done
