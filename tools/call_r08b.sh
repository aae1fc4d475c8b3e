# r08b: new oracle tests on the device; conv1 weight gradients batched 1 / 2 / 4 / 8 layers per launch on one box
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu -k "head_and_tail or criterion_kernels or wgrad_batch or recorded or record_refuses or paper_best or batch16" 2>&1 | tail -n 12 > gpurun_out/r08b_tests.txt
cat gpurun_out/r08b_tests.txt
for rep in 1 2; do for wb in 1 4 8 2; do
  SEPK_WGRAD_BATCH=$wb timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass --no-kernel-timing 2>gpurun_out/r08b_err_$wb.txt | tail -n 1 > gpurun_out/r08b_bench_wb$wb.json
  python -c "
import json; d=json.load(open('gpurun_out/r08b_bench_wb$wb.json')); print('wgrad batch $wb', round(d['ms_per_step'],3), 'ms', d['config'].get('final_loss'))" 2>&1 | tee -a gpurun_out/r08b_summary.txt
done; done
SEPK_WGRAD_BATCH=4 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stock --no-pmc --no-f32-pass 2>gpurun_out/r08b_err_kt.txt | tail -n 1 > gpurun_out/r08b_bench_kt.json
cp profiles/bench_detail.json gpurun_out/r08b_bench_detail.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r08b_bench_detail.json'))
bk=d['roofline_by_kernel']
for k,v in sorted(bk.items(), key=lambda kv:-kv[1]['ms_per_step'])[:14]:
    print(f"{k:45s} n={v['launches_per_step']:5.1f} avg={v['avg_us']:7.1f}us ms={v['ms_per_step']:6.3f} frac={v.get('hbm_frac',0):.3f}")
PY
