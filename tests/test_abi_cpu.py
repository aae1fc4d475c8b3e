"""CPU: the C-ABI boundary.  libsepkernels.so must load (no GPU needed for that) and export every entry point that
include/sepkernels.h declares; the ctypes mirror of the descriptor structs must have the C layout; and the product must
fail loudly -- not fall back -- when the library is missing or a CPU tensor reaches a kernel."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

import sepkernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sepkernels.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t|const char\s*\*)\s+(sep_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 29
    lib = sepkernels.load()
    for n in names:
        assert hasattr(lib, n), "{} declared in sepkernels.h but not exported".format(n)
        assert n in sepkernels.SIGNATURES, "{} has no ctypes signature".format(n)
    assert sorted(sepkernels.SIGNATURES) == names
    assert lib.sep_version() == sepkernels.ABI_VERSION == 22
    header = open(HEADER).read()
    assert "#define SEP_ABI_VERSION 22" in header and "#define SEP_STATS_SLOTS 16" in header
    assert sepkernels.STATS_SLOTS == 16


def test_descriptor_layouts_match_the_header():
    # both: 12 int32 + float (+4 pad), double at offset 56, then pointers: the layouts the header's field order implies on LP64
    assert sepkernels.GemmDesc.arith.offset == 44 and sepkernels.GemmDesc.eps.offset == 48
    assert sepkernels.GemmDesc.count.offset == 56 and sepkernels.GemmDesc.A.offset == 64
    assert ctypes.sizeof(sepkernels.GemmDesc) == 64 + 25 * 8
    assert sepkernels.GemmDesc.A_pk.offset == 64 + 23 * 8 and ctypes.sizeof(sepkernels.PackSeg) == 40
    assert sepkernels.WgradDesc.arith.offset == 44 and sepkernels.WgradDesc.count.offset == 56 and sepkernels.WgradDesc.G.offset == 64
    assert ctypes.sizeof(sepkernels.WgradDesc) == 64 + 10 * 8
    assert ctypes.sizeof(sepkernels.ReduceSeg) == 40


def test_argument_errors_come_back_through_the_abi():
    lib = sepkernels.load()
    d = sepkernels.GemmDesc(B=1, M=128, K=100, T=10, ldt=128)        # K not a multiple of the chunk depth
    assert lib.sep_pw_gemm(ctypes.byref(d), None) < 0
    assert b"K=100" in lib.sep_last_error()
    assert lib.sep_pw_gemm(None, None) < 0


def test_cpu_tensor_is_rejected_loudly():
    with pytest.raises(sepkernels.SepKernelsError):
        sepkernels.HipBackend().repack(torch.zeros(4, 8), 8, torch.zeros(4, 8), 8, 4, 8)


def test_missing_library_is_a_hard_error():
    code = ("import sys; sys.path.insert(0, {!r}); import sepkernels\n"
            "try:\n    sepkernels.load()\nexcept sepkernels.SepKernelsError as e:\n    print('LOUD', 'no CPU/PyTorch fallback' in str(e))\n"
            ).format(os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
    env = dict(os.environ, SEPKERNELS_LIB="/nonexistent/libsepkernels.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "LOUD True" in out.stdout, out.stdout + out.stderr


def test_integration_md_struct_stub_matches_the_binding():
    """INTEGRATION.md route B shows the ctypes struct a maintainer would copy: its field list must be the package's GemmDesc (a missing
    pointer shifts every field behind it)."""
    import re
    import sepkernels
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("class GemmDesc(ctypes.Structure)"):]
    block = block[:block.index("_lib.sep_pw_gemm.argtypes")]
    names = re.findall(r'"([A-Za-z_0-9]+)"', block)
    assert names == [n for n, _ in sepkernels.GemmDesc._fields_]
