#!/bin/bash
# NOT part of the test suite: the enabling procedure of the four-sequence LSTM sweeps (csrc/lstm.hip: lstm_fwd4_kernel /
# lstm_bwd4_kernel on v_mfma_f32_4x4x1_16b_f32), written without a GPU at hand and OFF by default (SEPK_LSTM_NS4 unset); their source has
# been executed on the host against the restatement (tests/test_kernel_source_on_host_cpu.py), what remains is the instruction's operand layout and timing.
# On an MI355X, from the repository root (about 40 s):
#   1. the operand layout the kernels assume            -> "layout: PASS"
#   2. the LSTM kernel tests and the DPRNN-TasNet / GALRNet / DPTNet model tests with the variant forced on
#   3. sweep timings, 16 against 4 sequences per workgroup, at the DPRNN-TasNet config-4 shapes
# If all three are good: make 2 ("when 16-sequence workgroups would leave compute units idle") the default in few_sequences().
set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma4x4_probe tools/mfma4x4_probe.hip 2>/dev/null
gpurun_out/mfma4x4_probe
SEPK_LSTM_NS4=1 python -m pytest tests/test_gpu_kernels.py -q -k lstm
SEPK_LSTM_NS4=1 python -m pytest tests/test_gpu_model.py -q -k "dprnn or galrnet or dptnet"
for mode in 0 1; do echo "SEPK_LSTM_NS4=$mode"; SEPK_LSTM_NS4=$mode python tools/lstm_bench.py; done
