# r04g: random shapes through the f16x3 weight gradient and the dense layers ON THE DEVICE (asynchronous hazards the host simulation cannot see)
export PYTHONPATH=dnn-based_source_separation_amd/src
mkdir -p gpurun_out
timeout 600 python tools/gpu_fuzz_wgrad.py 150 7 2>&1 | grep -v amdgpu | tee gpurun_out/r04g_fuzz_wgrad.txt
