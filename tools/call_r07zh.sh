# r07zh: clocks and power of the device while the headline step runs (rocm-smi polled beside bench.py)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 200 python bench.py --steps 2500 --warmup 5 --no-cpu-baseline --no-f32-pass --no-kernel-timing --no-pmc --no-stock > gpurun_out/r07zh_bench.log 2>&1 ) &
BP=$!
for i in $(seq 1 30); do sleep 2; echo -n "t=$((2*i))s "; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Current Socket" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; if ! kill -0 $BP 2>/dev/null; then break; fi; done | tee gpurun_out/r07zh_power.txt
wait $BP
tail -n 1 gpurun_out/r07zh_bench.log | cut -c1-200
