# r05y: register-fed weight-gradient kernel (pw_wgrad_rf16_kernel): parity tests, device fuzz against fp64, A/B against the raw-staging form
cd $GRAFT_REPO_ROOT
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | tail -3 )
( timeout 600 python tools/gpu_fuzz_wgrad.py 2>&1 | tail -4 )
for rf in 1 0 1 0; do echo "== SEPK_WGRAD_RF=$rf"; SEPK_WGRAD_RF=$rf timeout 300 python tools/gemm_bench.py --only W --reps 20 2>&1 | grep "^W" | cut -c1-110; done
