"""Development / test tool: the asm-free kernel files of csrc/ (stream.hip, cln.hip, loss.hip, lstm.hip) compiled as plain C++ against a
stand-in for the few pieces of the HIP programming model they use (tools/hostsim/include/hip/hip_runtime.h: one host thread per lane,
pthread barriers for workgroup and wave, shuffles and the MFMA instructions as wave collectives) into a shared library with the SAME
C ABI as libsepkernels.so, so that the kernels' own source can be run -- slowly -- on CPU tensors through the same Python binding.
One mechanical rewrite is applied to the copies that get compiled: `extern __shared__ T name[];` (dynamic LDS) becomes a pointer to a
per-launch buffer.  The GEMM files use inline assembly and LDS-DMA and are out of reach.

    from hostsim import build, HostSimBackend          (tests/test_kernel_source_on_host_cpu.py)
"""
import ctypes
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dnn-based_source_separation_amd", "csrc")
FILES = ("stream", "cln", "loss", "lstm")
_DYN = re.compile(r"extern __shared__ (?:__attribute__\(\(aligned\(\d+\)\)\) )?(\w+) (\w+)\[\];")


def compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c                                   # ext_vector_type needs clang
    return None


def build(workdir):
    """-> path of libsepkernels_hostsim.so, built in `workdir`"""
    cxx = compiler()
    if cxx is None:
        raise RuntimeError("hostsim needs clang++")
    inc = os.path.join(ROOT, "tools", "hostsim", "include")
    objs = []
    for f in FILES:
        src = open(os.path.join(CSRC, f + ".hip")).read()
        cpp = os.path.join(workdir, f + ".cpp")
        open(cpp, "w").write(_DYN.sub(r"\1* \2 = (\1*)sim_dynamic_lds();", src))
        objs.append(os.path.join(workdir, f + ".o"))
        subprocess.check_call([cxx, "-std=c++17", "-O1", "-fPIC", "-pthread", "-I", inc, "-I", CSRC, "-c", cpp, "-o", objs[-1]])
    objs.append(os.path.join(workdir, "sim_main.o"))
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-fPIC", "-pthread", "-I", inc, "-c", os.path.join(ROOT, "tools", "hostsim", "sim_main.cpp"), "-o", objs[-1]])
    so = os.path.join(workdir, "libsepkernels_hostsim.so")
    subprocess.check_call([cxx, "-shared", "-pthread", "-o", so] + objs)
    return so


class HostSimBackend:
    """sepkernels.HipBackend driving the host-simulation library on CPU tensors.  Swaps the binding's library handle, pointer check and
    stream getter for the lifetime of the object's `with` block -- test infrastructure, the product's own checks are untouched."""

    def __init__(self, so):
        self.so = so

    def __enter__(self):
        import sepkernels
        lib = ctypes.CDLL(self.so)
        for name, argtypes in sepkernels.SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:                         # the GEMM entry points do not exist here
                fn.argtypes = argtypes
                fn.restype = ctypes.c_char_p if name == "sep_last_error" else ctypes.c_int
        self._saved = (sepkernels._lib, sepkernels._ptr, sepkernels._stream)

        def ptr(t, dtype=None):
            if t is None:
                return None
            assert not t.is_cuda and t.is_contiguous() and (dtype is None or t.dtype == dtype), (t.device, t.dtype, dtype)
            return t.data_ptr()
        sepkernels._lib, sepkernels._ptr, sepkernels._stream = lib, ptr, (lambda: None)
        return sepkernels.HipBackend()

    def __exit__(self, *exc):
        import sepkernels
        sepkernels._lib, sepkernels._ptr, sepkernels._stream = self._saved
