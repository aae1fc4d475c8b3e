"""CPU: a HIP kernel file's OWN SOURCE executed on the host.  tools/lstm_hostsim.py compiles csrc/lstm.hip with a stand-in for the few
pieces of the HIP programming model it uses (one thread per lane, pthread barriers for the workgroup and for each wave, the MFMA
instructions as collective operations of a wave under their documented operand layouts) and compares sep_lstm_fwd / sep_lstm_bwd with the
step-by-step restatement of tests/emulator.py: the 16-sequence sweeps that run on the device, and the four-sequence sweeps
(SEPK_LSTM_NS4) that were written without a GPU at hand.  Catches indexing / synchronisation mistakes in the kernel source before any GPU
minute is spent; what it cannot see is timing and the hardware's operand layout itself (tools/mfma4x4_probe.hip)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not (os.path.exists(CLANG) or shutil.which("clang++")), reason="needs clang++ (ext_vector_type)")
def test_lstm_kernel_source_runs_on_the_host_and_matches_the_restatement():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lstm_hostsim.py")], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "agrees with the restatement" in r.stdout
    assert r.stdout.count("max |diff|") == 8           # 2 shapes on the 16-sequence kernels + 6 on the four-sequence ones
