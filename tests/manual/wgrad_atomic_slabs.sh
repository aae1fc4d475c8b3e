#!/bin/bash
# NOT part of the test suite: the measuring procedure of the one-slab weight-gradient accumulation (sep_wgrad_desc.accumulate,
# SEPK_WGRAD_ATOMIC=1; functionally checked on the host simulation, never timed).  On an MI355X, from the repository root (about 2 minutes):
#   1. the kernel cases on the device (all four weight-gradient kernels, three arithmetics)
#   2. the model-level golden tests with the switch on (gradient tolerances unchanged)
#   3. the headline step, switch off / on, in ONE call (box-to-box variance is +-4 %)
# Keep it if the step gets faster and the run-to-run variation of the gradients' last bits is acceptable for the recipes.
set -e
python -m pytest tests/test_gpu_kernels.py -q -k "accumulated_onto_one_slab"
SEPK_WGRAD_ATOMIC=1 python -m pytest tests/test_gpu_model.py -q -k "golden or oracle" -x
for mode in 0 1; do echo "SEPK_WGRAD_ATOMIC=$mode"; SEPK_WGRAD_ATOMIC=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-pass | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['roofline_wgrad']['avg_launch_ms'], 'ms per weight-gradient launch')"; done
