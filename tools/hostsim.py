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


def build(workdir, sanitize=None):
    """-> path of libsepkernels_hostsim.so, built in `workdir`; sanitize: None, "address" or "thread" (the process that loads a
    sanitized build must run with that sanitizer's runtime preloaded, see `python tools/hostsim.py --asan | --tsan`)"""
    cxx = compiler()
    if cxx is None:
        raise RuntimeError("hostsim needs clang++")
    inc = os.path.join(ROOT, "tools", "hostsim", "include")
    san = ["-g", "-fsanitize=" + sanitize, "-fno-omit-frame-pointer"] if sanitize else []
    objs = []
    for f in FILES:
        src = open(os.path.join(CSRC, f + ".hip")).read()
        cpp = os.path.join(workdir, f + ".cpp")
        open(cpp, "w").write(_DYN.sub(r"\1* \2 = (\1*)sim_dynamic_lds();", src))
        objs.append(os.path.join(workdir, f + ".o"))
        subprocess.check_call([cxx, "-std=c++17", "-O1", "-fPIC", "-pthread"] + san + ["-I", inc, "-I", CSRC, "-c", cpp, "-o", objs[-1]])
    objs.append(os.path.join(workdir, "sim_main.o"))
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-fPIC", "-pthread"] + san + ["-I", inc, "-c", os.path.join(ROOT, "tools", "hostsim", "sim_main.cpp"), "-o", objs[-1]])
    so = os.path.join(workdir, "libsepkernels_hostsim.so")
    subprocess.check_call([cxx, "-shared", "-pthread"] + (["-shared-libsan", "-fsanitize=" + sanitize] if sanitize else []) + ["-o", so] + objs)
    return so


def sanitizer_runtime(kind):
    """kind: "asan" or "tsan" -> path of the shared runtime next to the compiler, or None"""
    cxx = compiler()
    out = subprocess.check_output([cxx, "-print-file-name=libclang_rt.{}-x86_64.so".format(kind)], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


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


_SANITIZED_CHILD = r"""
import sys, time
sys.path[:0] = {paths!r}
import torch
torch.set_num_threads(1)
import hostsim
import test_gpu_kernels as GK
import test_kernel_source_on_host_cpu as H
with hostsim.HostSimBackend({so!r}) as K:
    GK.HIP, GK.to_device, GK.device_sync, GK.device_name = K, (lambda t: t.clone()), (lambda: None), (lambda: "cpu")
    for name, params in H.CASES:
        for p in params:
            t0 = time.time()
            getattr(GK, name)(*p)
            print("  {{:34s}} {{:28s}} {{:5.1f}} s".format(name, str(p), time.time() - t0), flush=True)
print("SANITIZED-RUN-COMPLETE")
"""


def main():
    """python tools/hostsim.py --asan | --tsan : the kernel cases of the CPU tier once more, with the kernel sources compiled under a
    sanitizer.  --asan: out-of-bounds reads / writes of global buffers (torch's allocations go through the intercepted allocator) and of
    the workgroup's LDS (function-local statics here) that happen to be harmless on the device.  --tsan: data races on LDS or global
    memory between the lanes of a workgroup -- a missing __syncthreads(), or code that silently relies on the lock-step of a wave (here
    every lane is a thread of its own, only barriers and the wave collectives order them)."""
    import sys
    import tempfile
    kind = "asan" if "--asan" in sys.argv else "tsan" if "--tsan" in sys.argv else None
    if kind is None:
        print(main.__doc__)
        return 0
    rt = sanitizer_runtime(kind)
    if rt is None:
        print("no", kind, "runtime next to", compiler())
        return 1
    marker = "ERROR: AddressSanitizer" if kind == "asan" else "WARNING: ThreadSanitizer"
    with tempfile.TemporaryDirectory() as d:
        so = build(d, sanitize="address" if kind == "asan" else "thread")
        paths = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), ROOT]
        env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=0", TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
        r = subprocess.run([sys.executable, "-c", _SANITIZED_CHILD.format(paths=paths, so=so)], env=env, capture_output=True, text=True)
    reports = r.stderr.count(marker)
    print(r.stdout[-6000:])
    if reports or "SANITIZED-RUN-COMPLETE" not in r.stdout:
        print(r.stderr[-8000:])
    print("{} reports: {}".format(marker.split(": ")[1], reports))
    return 1 if reports or r.returncode else 0


if __name__ == "__main__":
    raise SystemExit(main())
