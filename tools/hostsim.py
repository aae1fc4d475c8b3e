"""Development / test tool: every kernel file of csrc/ compiled as plain C++ against a stand-in for the pieces of the HIP programming
model they use (tools/hostsim/include/hip/hip_runtime.h: one host thread per lane, pthread barriers for workgroup and wave; shuffles,
DPP, ballots and the MFMA instructions as wave collectives; LDS-DMA as a wave-wide copy) into a shared library with the SAME C ABI as
libsepkernels.so, so that the kernels' own source can be run -- slowly -- on CPU tensors through the same Python binding.
Mechanical rewrites applied to the copies that get compiled (host_copy): `extern __shared__ T name[];` (dynamic LDS) becomes a pointer
to a per-launch buffer; empty asm statements (compiler fences with AMDGPU register constraints) and address-space attributes go; the
GEMM files' helpers whose body is inline assembly (LDS-DMA issue, v_max_f32 / v_max_f32_dpp / v_fma_mix_f32 one-liners) or rests on a
wave's lock-step (the flag stores of the producer / consumer protocol) get an equivalent C++ body (_HELPERS).

    from hostsim import build, HostSimBackend          (tests/test_kernel_source_on_host_cpu.py)
"""
import ctypes
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dnn-based_source_separation_amd", "csrc")
FILES = ("stream", "cln", "loss", "lstm", "linear", "attn", "rownorm", "gemm", "gemm_coop", "gemm_pc", "wgrad_pc", "wgrad_pc16", "sequence")
_DYN = re.compile(r"extern __shared__ (?:__attribute__\(\(aligned\(\d+\)\)\) )?(\w+) (\w+)\[\];")

# The GEMM files: helper functions whose bodies are inline assembly (or address-space casts) get a C++ body in the compiled copies.
_SPLIT2 = """static inline void {n}(const float x0, const float x1, unsigned& hi, unsigned& lo) {{
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(unsigned, h);
    _Float16 hh[2]; memcpy(hh, &hi, 4);
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0 - (float)hh[0], x1 - (float)hh[1]));      // v_fma_mix_f32: exact in fp32
}}
"""
_HELPERS = {
    "gemm_common.hpp": [
        ("glds16_asm", "static inline void glds16_asm(const float* base, unsigned voff, unsigned lds_dst) { sim_glds16((const char*)base + voff, lds_dst); }\n"),
        ("glds16_asm_v", "static inline void glds16_asm_v(const float* gsrc, unsigned lds_dst) { sim_glds16(gsrc, lds_dst); }\n"),
        ("lds_addr", "static inline unsigned lds_addr(const float* p) { return sim_lds_addr(p); }\n")],
    "gemm_pc.hip": [
        # the flag protocol between producer and consumer waves rests on the LDS executing ONE WAVE's accesses in order (every lane's panel
        # write precedes the wave's flag store).  Lanes are threads here: the wave meets before its flag store, and the flags are atomics.
        ("pcd_load4", "static inline pcd_i4 pcd_load4(const int* c) { pcd_i4 v; for (int i = 0; i < 4; ++i) v[i] = __atomic_load_n(c + i, __ATOMIC_ACQUIRE); return v; }\n"),
        ("pcd_post", "static inline void pcd_post(int* c, const int value) { sim_wave_sync(); __atomic_store_n(c, value, __ATOMIC_RELEASE); }\n"),
        ("pcd_vmax", "static inline float pcd_vmax(const float a, const float b) { return a > b ? a : b; }\n"),
        ("pcd_max_quad_neighbour", "static inline float pcd_max_quad_neighbour(const float m) { const float o = sim_read_lane(m, (threadIdx.x & 63) ^ 1); return m > o ? m : o; }\n"),
        ("pcd_split2_pair", _SPLIT2.format(n="pcd_split2_pair"))],
    "wgrad_pc16.hip": [
        ("w16_max_halves", "static inline float w16_max_halves(const float m) { const float o = sim_read_lane(m, (threadIdx.x & 63) ^ 32); return m > o ? m : o; }\n"),
        ("w16_max_neighbour", "static inline float w16_max_neighbour(const float m) { const float o = sim_read_lane(m, (threadIdx.x & 63) ^ 1); return m > o ? m : o; }\n"),
        ("w16_split2_pair", _SPLIT2.format(n="w16_split2_pair")),
        ("w16_dma_pieces8", "static inline void w16_dma_pieces8(const float* pa, const float* pb, const unsigned (&voffc)[8], unsigned lds1) { for (int q = 0; q < 8; ++q) "
                            "sim_glds16((const char*)(q < 4 ? pa : pb) + voffc[q] - ((q & 1) ? 0 : 4096), lds1 + 4096 * (q - 1)); }\n"),
        ("w16_dma_pieces4", "static inline void w16_dma_pieces4(const float* pa, const unsigned (&voffc)[4], unsigned lds1) { for (int q = 0; q < 4; ++q) "
                            "sim_glds16((const char*)pa + voffc[q] - ((q & 1) ? 0 : 4096), lds1 + 4096 * (q - 1)); }\n")],
    "gemm_coop.hip": [
        ("co_vmax", "static inline float co_vmax(const float a, const float b) { return a > b ? a : b; }\n"),
        ("co_quad_max", "static inline float co_quad_max(const float m) { float o = sim_read_lane(m, (threadIdx.x & 63) ^ 1); const float t = m > o ? m : o; "
                        "o = sim_read_lane(t, (threadIdx.x & 63) ^ 2); return t > o ? t : o; }\n"),
        ("co_split2_pair", _SPLIT2.format(n="co_split2_pair"))],
}


def host_copy(fname):
    """the text of csrc/<fname> as it is compiled for the host: dynamic LDS declarations, empty asm statements (compiler fences with
    AMDGPU register constraints), address-space attributes and the helpers of _HELPERS are rewritten; nothing else"""
    src = open(os.path.join(CSRC, fname)).read()
    for name, body in _HELPERS.get(fname, []):
        m = re.search(r"^__device__ __forceinline__ [\w\s\*]*\b%s\(" % re.escape(name), src, re.M)
        assert m, (fname, name)
        end = src.index("\n}\n", m.start()) + 3
        src = src[:m.start()] + body + src[end:]
    src = _DYN.sub(r"\1* \2 = (\1*)sim_dynamic_lds();", src)
    src = re.sub(r'asm volatile\(""[^;]*\);', ";", src)
    src = re.sub(r"__attribute__\(\(address_space\(\d\)\)\)[ \t]*", "", src)
    src = re.sub(r"\*\(const PCD_LDS int\*\)\((be_addr[^;]*)\);", r"*(const int*)sim_lds_ptr(\1);", src)      # an LDS BYTE ADDRESS used as a pointer
    # the two lanes holding the row halves of a column store the SAME exponent word: one instruction on the device, two racing threads here
    src = src.replace("sm.be[pb][colb + 128 * cc] = bexp[cc];", "__atomic_store_n(&sm.be[pb][colb + 128 * cc], bexp[cc], __ATOMIC_RELAXED);")
    src = src.replace("sm.be[pb][col_s] = bexp;", "__atomic_store_n(&sm.be[pb][col_s], bexp, __ATOMIC_RELAXED);")             # gemm_coop.hip: the four lanes of a quad
    return src


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
    open(os.path.join(workdir, "gemm_common.hpp"), "w").write(host_copy("gemm_common.hpp"))      # found before csrc/'s by the quoted includes
    procs = []
    for f in FILES:
        cpp = os.path.join(workdir, f + ".cpp")
        open(cpp, "w").write(host_copy(f + ".hip"))
        objs.append(os.path.join(workdir, f + ".o"))
        procs.append(subprocess.Popen([cxx, "-std=c++17", "-O1", "-fPIC", "-pthread"] + san + ["-I", workdir, "-I", inc, "-I", CSRC, "-c", cpp, "-o", objs[-1]]))
    for pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("hostsim: compiling a kernel file for the host failed")
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
                fn.restype = sepkernels._RESTYPES.get(name, ctypes.c_int)
        self._saved = (sepkernels._lib, sepkernels._ptr, sepkernels._stream)

        def ptr(t, dtype=None):
            if t is None:
                return None
            assert not t.is_cuda and t.is_contiguous() and (dtype is None or t.dtype == dtype), (t.device, t.dtype, dtype)
            sepkernels._keep(t)                        # (a Sequence being recorded keeps what its ops point at)
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
    import os
    only = [o for o in os.environ.get("HOSTSIM_ONLY", "").split(",") if o]        # python tools/hostsim.py --tsan --only rownorm,relu_drop,attention
    for name, params in H.CASES:
        if only and not any(o in name for o in only):
            continue
        for p in params:
            t0 = time.time()
            getattr(GK, name)(*p)
            print("  {{:34s}} {{:28s}} {{:5.1f}} s".format(name, str(p), time.time() - t0), flush=True)
    import sepkernels
    for arith, name, args in H.GEMM_CASES:
        if only and not any(o in name for o in only):
            continue
        t0 = time.time()
        if arith is None:
            getattr(GK, name)(*args)
        else:
            prev = sepkernels.set_gemm_arith(arith.split("-")[0]); GK.PACKED[0] = arith.endswith("packed")
            getattr(GK, name)(*args, arith)
            GK.PACKED[0] = False; sepkernels.set_gemm_arith(prev)
        print("  {{:12s}} {{:34s}} {{:24s}} {{:5.1f}} s".format(str(arith), name[:34], str(args), time.time() - t0), flush=True)
print("SANITIZED-RUN-COMPLETE")
"""


def run_case(argv):
    """python tools/hostsim.py --run <test function of tests/test_gpu_kernels.py> [its arguments ...] [--arith f32|f16x3|f16x3-packed|bf16x6]
    e.g.  --run test_gemm_packed_weights_model_shapes 128 512 130      --run test_wgrad_plain 2 256 128 300 3 --arith f16x3
    One kernel case on the host simulation of the CURRENT sources (rebuilt on every call, ~20 s): the pre-check of a kernel edit."""
    import ast
    import sys
    import tempfile
    import time
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    arith = None
    if "--arith" in argv:
        i = argv.index("--arith")
        arith, argv = argv[i + 1], argv[:i] + argv[i + 2:]
    name, args = argv[0], []
    for a in argv[1:]:
        try:
            args.append(ast.literal_eval(a))
        except (ValueError, SyntaxError):
            args.append(a)
    import sepkernels
    import test_gpu_kernels as GK
    with tempfile.TemporaryDirectory() as d:
        so = build(d)
        with HostSimBackend(so) as K:
            GK.HIP, GK.to_device, GK.device_sync, GK.device_name = K, (lambda t: t.clone()), (lambda: None), (lambda: "cpu")
            if arith:
                sepkernels.set_gemm_arith(arith.split("-")[0])
                GK.PACKED[0] = arith.endswith("packed")
                args.append(arith)
            t0 = time.time()
            getattr(GK, name)(*args)
            print("{}{} on the host simulation: ok, {:.1f} s".format(name, tuple(args), time.time() - t0))
    return 0


def main():
    """python tools/hostsim.py --asan | --tsan : the kernel cases of the CPU tier once more, with the kernel sources compiled under a
    sanitizer.  --asan: out-of-bounds reads / writes of global buffers (torch's allocations go through the intercepted allocator) and of
    the workgroup's LDS (function-local statics here) that happen to be harmless on the device.  --tsan: data races on LDS or global
    memory between the lanes of a workgroup -- a missing __syncthreads(), or code that silently relies on the lock-step of a wave (here
    every lane is a thread of its own, only barriers and the wave collectives order them)."""
    import sys
    import tempfile
    if "--run" in sys.argv:
        return run_case(sys.argv[sys.argv.index("--run") + 1:])
    kind = "asan" if "--asan" in sys.argv else "tsan" if "--tsan" in sys.argv else None
    if kind is None:
        print(main.__doc__)
        print(run_case.__doc__)
        return 0
    rt = sanitizer_runtime(kind)
    if rt is None:
        print("no", kind, "runtime next to", compiler())
        return 1
    marker = "ERROR: AddressSanitizer" if kind == "asan" else "WARNING: ThreadSanitizer"
    with tempfile.TemporaryDirectory() as d:
        so = build(d, sanitize="address" if kind == "asan" else "thread")
        paths = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), ROOT]
        if "--only" in sys.argv:
            os.environ["HOSTSIM_ONLY"] = sys.argv[sys.argv.index("--only") + 1]
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
