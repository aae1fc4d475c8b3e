"""Development tool: csrc/lstm.hip's OWN SOURCE executed on the host -- one thread per lane, barriers for the workgroup and for each
wave, the two MFMA instructions as collective operations under the operand layouts the kernels assume (tools/lstm_hostsim/common.hpp) --
and compared with the step-by-step restatement the kernel tests use (tests/emulator.py), for the 16-sequence sweeps (a check of the
simulator itself: those run on the device every day) and for the four-sequence sweeps prepared without a GPU at hand (SEPK_LSTM_NS4=1).
Complements tools/lstm4_model.py (a Python transcription): here nothing is transcribed.

    python tools/lstm_hostsim.py
"""
import ctypes
import os
import shutil
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src")]
from emulator import EmuBackend          # noqa: E402


def build(workdir):
    sim = os.path.join(ROOT, "tools", "lstm_hostsim")
    for f in ("common.hpp", "main.cpp"):
        shutil.copy(os.path.join(sim, f), workdir)
    shutil.copy(os.path.join(ROOT, "dnn-based_source_separation_amd", "csrc", "lstm.hip"), workdir)      # next to the stand-in common.hpp
    so = os.path.join(workdir, "liblstm_hostsim.so")
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "clang++"      # ext_vector_type
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-x", "c++", os.path.join(workdir, "main.cpp"), "-o", so],
                          cwd=workdir)
    return so


def run(so, mode, cases):
    """in a child process: SEPK_LSTM_NS4 is read once per process"""
    code = r'''
import ctypes, sys, torch
sys.path[:0] = {paths!r}
from emulator import EmuBackend
lib = ctypes.CDLL({so!r})
vp, I = ctypes.c_void_p, ctypes.c_int
lib.sep_lstm_fwd.argtypes = [vp] * 5 + [I] * 4 + [vp]
lib.sep_lstm_bwd.argtypes = [vp] * 5 + [I] * 4 + [vp]
E = EmuBackend()
torch.manual_seed(0)
worst = 0.0
for H, nseq, L, reverse in {cases!r}:
    nd = 2 if reverse == 2 else 1
    shape = lambda n: (nd, nseq, L, n) if nd == 2 else (nseq, L, n)
    xg = torch.randn(*shape(4 * H))
    whh = (H ** -0.5) * torch.randn(*((nd, 4 * H, H) if nd == 2 else (4 * H, H)))
    want = [torch.empty(*shape(n)) for n in (H, 4 * H, H)]
    E.lstm_fwd(xg, whh, *want, nseq, L, H, reverse)
    got = [torch.full(shape(n), float("nan")) for n in (H, 4 * H, H)]
    assert lib.sep_lstm_fwd(xg.data_ptr(), whh.data_ptr(), *[g.data_ptr() for g in got], nseq, L, H, reverse, None) == 0
    dho = torch.randn(*shape(H))
    dwant = torch.empty(*shape(4 * H))
    E.lstm_bwd(dho, want[1], want[2], whh, dwant, nseq, L, H, reverse)
    dgot = torch.full(shape(4 * H), float("nan"))
    assert lib.sep_lstm_bwd(dho.data_ptr(), want[1].data_ptr(), want[2].data_ptr(), whh.data_ptr(), dgot.data_ptr(), nseq, L, H, reverse, None) == 0
    errs = [(a - b).abs().max().item() for a, b in zip(want + [dwant], got + [dgot])]
    assert all(e == e for e in errs), "an output element was never written"
    worst = max(worst, *errs)
    print("  H={{:3d}} nseq={{:2d}} L={{}} reverse={{}}: max |diff| h {{:.1e}} gates {{:.1e}} c {{:.1e}} dxg {{:.1e}}".format(H, nseq, L, reverse, *errs), flush=True)
assert worst < 2e-5, worst
'''.format(paths=[os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src")], so=so, cases=cases)
    env = dict(os.environ, SEPK_LSTM_NS4=str(mode))
    subprocess.check_call([sys.executable, "-c", code], env=env)


def main():
    with tempfile.TemporaryDirectory() as d:
        so = build(d)
        print("16 sequences per workgroup (the kernels in use; checks the simulator):")
        run(so, 0, [(16, 5, 3, 0), (32, 18, 2, 1)])
        print("4 sequences per workgroup (SEPK_LSTM_NS4=1):")
        run(so, 1, [(16, 5, 4, 0), (16, 6, 3, 1), (32, 9, 3, 0), (32, 3, 2, 2), (64, 5, 2, 1), (128, 6, 2, 2)])
    print("csrc/lstm.hip executed on the host agrees with the restatement")


if __name__ == "__main__":
    main()
