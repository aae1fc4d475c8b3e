"""
sepkernels -- ctypes binding of libsepkernels.so (include/sepkernels.h), the MI355X/gfx950 kernels of the
Conv-TasNet separation path.

This package is plumbing only: it marshals torch tensors (device memory + the current HIP stream) into the
plain-pointer C ABI.  There is NO CPU implementation behind it: every entry point raises if the shared
library is missing or if it is handed a non-GPU tensor.
"""
import ctypes
import os

import torch  # imported first on purpose: libsepkernels must bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEPKERNELS_LIB", os.path.normpath(os.path.join(_HERE, "..", "..", "libsepkernels.so")))

# ---- constants mirrored from include/sepkernels.h -------------------------------------------------
PRO_NONE, PRO_PRELU, PRO_GLN, PRO_GLN_PRELU, PRO_GLN_BWD = 0, 1, 2, 3, 4
EPI_STATS_PRELU, EPI_RESIDUAL, EPI_SIGMOID, EPI_PRELU_BWD, EPI_ROWSUMS, EPI_ROWSUMS_PRELU = 1, 2, 4, 8, 16, 32
ABI_VERSION = 23
LSTM_INTERLEAVED = 0x400       # sep_lstm_fwd / sep_lstm_bwd with reverse = 2: h_out / dh_out as one (nseq, L, 2H) buffer
STATS_SLOTS = 16   # SEP_STATS_SLOTS: gLN statistics are double[B][STATS_SLOTS][2]
ARRIVE_INTS = 17   # SEP_ARRIVE_INTS: arrival counters of the gLN-backward publishers, int[B][ARRIVE_INTS]
ARITH_F32, ARITH_BF16X6, ARITH_F16X3 = 0, 1, 2     # SEP_ARITH_*: how sep_pw_gemm forms its fp32 products (include/sepkernels.h)
_ARITH_NAMES = {"f32": ARITH_F32, "bf16x6": ARITH_BF16X6, "f16x3": ARITH_F16X3}
_gemm_arith = None


def gemm_arith():
    """Arithmetic requested of sep_pw_gemm / sep_pw_wgrad when a call does not name one:
    SEPK_GEMM_ARITH = f16x3 (default) | bf16x6 | f32."""
    global _gemm_arith
    if _gemm_arith is None:
        name = os.environ.get("SEPK_GEMM_ARITH", "f16x3")
        if name not in _ARITH_NAMES:
            raise SepKernelsError("SEPK_GEMM_ARITH must be one of {} (got '{}')".format(sorted(_ARITH_NAMES), name))
        _gemm_arith = _ARITH_NAMES[name]
    return _gemm_arith


_weights_amax = None


def set_weights_amax(t):
    """Device scalar >= max|A| over every weight the following sep_pw_gemm calls will use (SEP_ARITH_F16X3), or None.
    Returns the previous one.  net.forward / net.backward set it once per pass from the co-located parameter buffer."""
    global _weights_amax
    prev, _weights_amax = _weights_amax, t
    return prev


def gemm_arith_name():
    return {v: k for k, v in _ARITH_NAMES.items()}[gemm_arith()]


def arith_code(name):
    return _ARITH_NAMES[name]


def set_gemm_arith(name):
    """'f32' | 'bf16x6' | 'f16x3'; returns the previous setting's name."""
    global _gemm_arith
    prev = gemm_arith_name()
    _gemm_arith = _ARITH_NAMES[name]
    return prev

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32


class GemmDesc(ctypes.Structure):
    _fields_ = [(n, _i32) for n in ("B", "M", "K", "T", "ldt", "trans_a", "k_split", "m_split", "pro_mode", "epi_flags",
                                    "accumulate", "arith")] + [("eps", ctypes.c_float), ("count", ctypes.c_double)] + \
               [(n, _vp) for n in ("A", "A2", "X", "X2", "Y", "Y2", "bias", "pro_alpha", "pro_stats", "pro_gamma", "pro_beta",
                                   "pro_aux", "pro_bsum", "pro_bacc", "pro_store", "pro_dalpha", "epi_alpha", "epi_stats", "epi_res",
                                   "epi_aux", "epi_dalpha", "epi_rowpart", "a_amax", "A_pk", "a_rscale")]


class WgradDesc(ctypes.Structure):
    _fields_ = [(n, _i32) for n in ("B", "M", "N", "T", "ldt", "g_split", "g_mul", "g_div", "x_mode", "x_div", "nsplit", "arith")] + \
               [("eps", ctypes.c_float), ("count", ctypes.c_double)] + \
               [(n, _vp) for n in ("G", "G2", "Gaux", "X", "x_alpha", "x_stats", "x_gamma", "x_beta", "partial", "partial_bias", "G2_pre", "g2_exps", "g2_sums")]


class PackSeg(ctypes.Structure):
    _fields_ = [("W", _vp), ("dst", _vp), ("rscale", _vp), ("M", _i32), ("K", _i32), ("trans", _i32), ("ldw", _i32)]


class PackedA:
    """One weight matrix as written by sep_pack_weights: A ([M][K], in the orientation of the product) as {hi, lo} fp16
    groups (`data`, M*K fp32-sized words) plus the inverse row scales (`rscale`, [M]).  `src` remembers what it was packed
    from (the CPU emulator of the tests multiplies with that)."""
    __slots__ = ("data", "rscale", "M", "K", "src")

    def __init__(self, data, rscale, M, K, src=None):
        self.data, self.rscale, self.M, self.K, self.src = data, rscale, M, K, src


class FinalizeSeg(ctypes.Structure):
    _fields_ = [(n, _vp) for n in ("rowpart", "stats", "gamma", "bsum", "pbeta", "pgamma", "pextra")] + [("count", ctypes.c_double), ("eps", ctypes.c_float)] + \
               [(n, _i32) for n in ("ntile", "nq", "B", "C")]


class ReduceSeg(ctypes.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("n", _i32), ("nslab", _i32), ("stride", ctypes.c_int64),
                ("accumulate", _i32), ("scale", ctypes.c_float)]


SEQ_MAX_ARGS = 26


class SeqArg(ctypes.Union):
    _fields_ = [("i", ctypes.c_int64), ("f", ctypes.c_double), ("p", _vp)]


class SeqOp(ctypes.Structure):
    _fields_ = [("fn", _i32), ("nargs", _i32), ("args", SeqArg * SEQ_MAX_ARGS)]


# name -> argtypes (restype is always int unless noted); also the list the symbol-export test checks
_I, _F, _D, _L = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_int64
SIGNATURES = {
    "sep_version": [],
    "sep_last_error": [],
    "sep_pw_gemm": [ctypes.POINTER(GemmDesc), _vp],
    "sep_pack_weights": [ctypes.POINTER(PackSeg), _I, _vp],
    "sep_pw_wgrad": [ctypes.POINTER(WgradDesc), _vp],
    "sep_pw_wgrad_batch": [ctypes.POINTER(WgradDesc), _I, _vp],
    "sep_reduce_slabs": [ctypes.POINTER(ReduceSeg), _I, _vp],
    "sep_f64_to_f32": [_vp, _vp, _I, _I, _vp],
    "sep_encoder_fwd": [_vp, _vp, _vp, _vp] + [_I] * 10 + [_vp],
    "sep_unfold": [_vp, _vp] + [_I] * 8 + [_vp],
    "sep_dwconv_fwd": [_vp] * 10 + [_I] * 5 + [_F, _vp],
    "sep_dwconv_bwd": [_vp] * 18 + [_I] * 5 + [_F, _vp],
    "sep_gln_bwd_finalize": [_vp, _I, _I, _vp, _vp, _D, _F, _vp, _vp, _vp, _vp, _I, _I, _vp],
    "sep_gln_bwd_finalize_batch": [ctypes.POINTER(FinalizeSeg), _I, _vp],
    "sep_gln_bwd_from_wgrad": [_vp] * 6 + [_D, _F] + [_vp] * 6 + [_I] * 6 + [_vp],
    "sep_head_bwd": [_vp] * 6 + [_I] * 4 + [_D, _F, _I, _vp],
    "sep_decoder_fwd": [_vp] * 5 + [_I] * 10 + [_vp],
    "sep_decoder_bwd": [_vp] * 6 + [_I] * 11 + [_vp],
    "sep_softmax_ch_fwd": [_vp] + [_I] * 4 + [_vp],
    "sep_softmax_ch_bwd": [_vp, _vp] + [_I] * 4 + [_vp],
    "sep_cln_ws_bytes": [_I] * 4,                                    # returns size_t
    "sep_cln_fwd": [_vp] * 7 + [_I] * 4 + [_F, _vp, _vp],
    "sep_cln_bwd": [_vp] * 9 + [_I] * 4 + [_F, _vp, _vp, _vp],
    "sep_attn_fwd": [_vp] * 3 + [_I] * 4 + [_F, _F, ctypes.c_ulonglong, _vp],
    "sep_attn_bwd": [_vp] * 6 + [_I] * 4 + [_F, _F, ctypes.c_ulonglong, _vp],
    "sep_gln_tokens_ws_bytes": [_I] * 3,                             # returns size_t
    "sep_gln_tokens_fwd": [_vp] * 6 + [_I] * 3 + [_F, _vp],
    "sep_gln_tokens_bwd": [_vp] * 7 + [_I] * 3 + [_vp],
    "sep_rownorm_parts": [_L, _I],
    "sep_rownorm_fwd": [_vp] * 7 + [_L, _I, _F, _F, ctypes.c_ulonglong, _vp],
    "sep_rownorm_bwd": [_vp] * 7 + [_L, _I, _F, ctypes.c_ulonglong, _vp],
    "sep_relu_drop_fwd": [_vp, _vp, _L, _F, ctypes.c_ulonglong, _vp],
    "sep_relu_drop_bwd": [_vp, _vp, _vp, _L, _F, _vp],
    "sep_gln_stats": [_vp, _vp, _I, _I, _I, _I, _vp],
    "sep_gln_apply": [_vp] * 5 + [_I] * 4 + [_D, _F, _vp],
    "sep_gln_bwd_rowsums": [_vp, _vp, _vp, _I, _I, _I, _I, _vp],
    "sep_gln_bwd_apply": [_vp] * 6 + [_I] * 4 + [_D, _F, _vp],
    "sep_repack": [_vp, _I, _vp, _I, _I, _I, _vp],
    "sep_segment": [_vp, _vp] + [_I] * 7 + [_vp],
    "sep_depthwise_fwd": [_vp] * 4 + [_I] * 8 + [_vp],
    "sep_depthwise_bwd_input": [_vp] * 3 + [_I] * 8 + [_vp],
    "sep_depthwise_bwd_weight": [_vp] * 3 + [_I] * 8 + [_vp],
    "sep_overlap_add": [_vp, _vp] + [_I] * 7 + [_vp],
    "sep_sisdr_dots": [_vp] * 5 + [_I] * 4 + [_vp],
    "sep_sisdr_from_dots": [_vp] * 4 + [_I] * 3 + [_F, _vp],
    "sep_sisdr_bwd": [_vp] * 7 + [_I] * 4 + [_F, _vp],
    "sep_pit_search": [_vp, _vp, _I, _I, _I, _I, _I, _vp, _vp, _vp],
    "sep_sinkhorn_fwd": [_vp] * 4 + [_I, _I, _F, _I, _vp],
    "sep_sinkhorn_bwd": [_vp] * 4 + [_I, _I, _F, _I, _vp],
    "sep_rowdiff_sums": [_vp, _vp, _vp, _L, _I, _vp],
    "sep_rowdiff_bwd": [_vp, _vp, _vp, _vp, _vp, _L, _I, _vp],
    "sep_sqnorm": [_vp, _vp, _L, _vp],
    "sep_adam_step": [_vp] * 5 + [_L] + [_F] * 7 + [_I, _vp],
    "sep_adam_step_dev": [_vp] * 5 + [_L, _vp, _vp] + [_F] * 6 + [_vp],
    "sep_lstm_fwd": [_vp] * 5 + [_I] * 4 + [_vp],
    "sep_lstm_bwd": [_vp] * 5 + [_I] * 4 + [_vp],
    "sep_linear_fwd": [_vp] * 5 + [_L, _I, _I, _vp],
    "sep_linear_bwd_input": [_vp] * 3 + [_L, _I, _I, _I, _vp],
    "sep_linear_bwd_weight": [_vp, _vp, _L, _vp, _vp, _L] + [_I] * 5 + [_vp],
    "sep_chunk_to_tokens": [_vp, _vp] + [_I] * 5 + [_vp],
    "sep_tokens_to_chunk": [_vp, _vp] + [_I] * 5 + [_vp],
    "sep_seq_count": [],
    "sep_seq_lookup": [ctypes.c_char_p],
    "sep_seq_name": [_I],                                            # returns const char*
    "sep_seq_nargs": [_I],
    "sep_run_sequence": [ctypes.POINTER(SeqOp), _I, _vp],
    "sep_memset": [_vp, _I, ctypes.c_size_t, _vp],
    "sep_absmax": [_vp, _L, _vp, _vp],
    "sep_pit_finish": [_vp, _vp, _vp, _I, _I, _I, _F, _F, _vp, _vp, _vp, _vp],
    "sep_axpby": [_vp, _F, _vp, _F, _vp, _L, _vp],
    "sep_split_rows": [_vp, _vp, _vp, _vp, _I, _I, _I, _I, _I, _vp],
}
_RESTYPES = {"sep_last_error": ctypes.c_char_p, "sep_seq_name": ctypes.c_char_p, "sep_cln_ws_bytes": ctypes.c_size_t,
             "sep_gln_tokens_ws_bytes": ctypes.c_size_t}

_lib = None


class SepKernelsError(RuntimeError):
    pass


def load():
    """Load libsepkernels.so (once).  Fails loudly -- there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib if _recording is None else _RecordingLib(_lib, _recording)
    if not os.path.exists(LIB_PATH):
        raise SepKernelsError(
            "libsepkernels.so not found at {} -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The HIP extension is mandatory; there is no CPU/PyTorch fallback.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    if lib.sep_version() != ABI_VERSION:
        raise SepKernelsError("libsepkernels ABI {} != binding ABI {}".format(lib.sep_version(), ABI_VERSION))
    _lib = lib
    return lib if _recording is None else _RecordingLib(lib, _recording)


# ---- recorded launch sequences (include/sepkernels.h, ABI 23) --------------------------------------------------------------------------
_recording = None          # the Sequence being recorded (every successful launch through load() is appended to it), or None


class Sequence:
    """The launches of one pass, recorded once and replayed by ONE call of sep_run_sequence (a C loop over the same entry points; no
    hipGraph).  Keeps alive everything the ops point at: the device tensors that went through the binding while recording (their
    addresses are in the ops) and the host-side descriptors / segment arrays."""

    def __init__(self):
        self.ops = []           # (name, fn id, [(kind, value)])
        self.keep = []
        self._arr = None

    def __len__(self):
        return len(self.ops)

    def add(self, lib, name, args):
        types = SIGNATURES[name][:-1]
        if len(types) > SEQ_MAX_ARGS:
            raise SepKernelsError("{} has {} arguments, a recorded op holds {}".format(name, len(types), SEQ_MAX_ARGS))
        fn = lib.sep_seq_lookup(name.encode())
        if fn < 0:
            raise SepKernelsError("{} cannot be recorded into a sequence".format(name))
        conv = []
        for ty, a in zip(types, args):
            if ty is _vp:
                conv.append(("p", int(a) if a else 0))
            elif isinstance(ty, type) and issubclass(ty, ctypes._Pointer):
                obj = getattr(a, "_obj", a)                # ctypes.byref(struct) -> the struct; an array instance is passed as it is
                self.keep.append(obj)
                conv.append(("p", ctypes.addressof(obj)))
            elif ty in (_F, _D):
                conv.append(("f", float(a)))
            else:
                conv.append(("i", int(a)))
        self.ops.append((name, fn, conv))
        self._arr = None

    def names(self):
        return [o[0] for o in self.ops]

    def _array(self):
        if self._arr is None:
            arr = (SeqOp * max(1, len(self.ops)))()
            for k, (_, fn, conv) in enumerate(self.ops):
                arr[k].fn, arr[k].nargs = fn, len(conv)
                for q, (kind, v) in enumerate(conv):
                    setattr(arr[k].args[q], kind, v)
            self._arr = arr
        return self._arr

    def run(self, first=0, last=None):
        """replay ops [first, last) on the current stream"""
        last = len(self.ops) if last is None else last
        if last <= first:
            return
        arr = self._array()
        base = ctypes.cast(ctypes.byref(arr, first * ctypes.sizeof(SeqOp)), ctypes.POINTER(SeqOp))
        lib = _lib if _lib is not None else load()
        _check(lib.sep_run_sequence(base, last - first, _stream()), "sep_run_sequence")


class _RecordingLib:
    """what load() hands out while a Sequence is being recorded: every entry point runs as usual and, when it succeeded and takes a
    stream, is appended to the sequence"""

    def __init__(self, lib, seq):
        self._lib, self._seq = lib, seq

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        sig = SIGNATURES.get(name)
        if not sig or sig[-1] is not _vp or name in ("sep_run_sequence",) or name in _RESTYPES:
            return fn
        lib, seq = self._lib, self._seq

        def call(*args):
            rc = fn(*args)
            if rc == 0:
                seq.add(lib, name, args)
            return rc
        return call


class recording:
    """with sepkernels.recording(seq): ... -- the launches issued inside go into `seq` (and are executed).  Not re-entrant."""

    def __init__(self, seq):
        self.seq = seq

    def __enter__(self):
        global _recording
        if _recording is not None:
            raise SepKernelsError("a sequence is already being recorded")
        _recording = self.seq
        return self.seq

    def __exit__(self, *exc):
        global _recording
        _recording = None
        return False


def is_recording():
    return _recording is not None


def _keep(*tensors):
    if _recording is not None:
        _recording.keep.extend(t for t in tensors if t is not None)


def _ptr(t, dtype=None):
    if t is None:
        return None
    if _recording is not None:
        _recording.keep.append(t)
    if not t.is_cuda:
        raise SepKernelsError("sepkernels is HIP-only: got a {} tensor (no CPU fallback exists)".format(t.device))
    if dtype is not None and t.dtype != dtype:
        raise SepKernelsError("expected dtype {}, got {}".format(dtype, t.dtype))
    if not t.is_contiguous():
        raise SepKernelsError("non-contiguous tensor passed to a kernel")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current HIP stream's handle.  torch.cuda.current_stream() builds a Stream object per call (9 us of the ~30 us a launch costs on
    the host: tools/host_profile.py, profiles/r07v_host_galrnet.txt); the raw accessor it rests on is 20 times cheaper."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _check(rc, name):
    if rc != 0:
        raise SepKernelsError("{} failed ({}): {}".format(name, rc, load().sep_last_error().decode()))


_f32, _f64 = torch.float32, torch.float64


class HipBackend:
    """Tensor-level facade over the C ABI.  tests/emulator.py implements the same interface with CPU torch
    arithmetic so that the host-side orchestration can be exercised without a GPU (test infrastructure only)."""

    name = "hip"

    def pw_gemm(self, *, B, M, K, T, ldt, A, X, Y, trans_a=0, A2=None, X2=None, k_split=0, Y2=None, m_split=0,
                pro_mode=PRO_NONE, epi_flags=0, accumulate=0, eps=1e-12, count=0.0, bias=None, pro_alpha=None,
                pro_stats=None, pro_gamma=None, pro_beta=None, pro_aux=None, pro_bsum=None, pro_bacc=None, pro_store=None,
                pro_dalpha=None, epi_alpha=None, epi_stats=None, epi_res=None, epi_aux=None, epi_dalpha=None,
                epi_rowpart=None, arith=None, a_amax=None, A_pk=None):
        arith = gemm_arith() if arith is None else arith
        if A_pk is not None and (A_pk.M != M or A_pk.K != K):
            raise SepKernelsError("packed weights are {}x{}, the product needs {}x{}".format(A_pk.M, A_pk.K, M, K))
        if arith == ARITH_F16X3 and a_amax is None:
            a_amax = _weights_amax
            if a_amax is None:      # stand-alone caller: the bound is formed here (two small kernels per call)
                a_amax = A.detach().abs().amax().reshape(1) if A2 is None else torch.maximum(A.detach().abs().amax(), A2.detach().abs().amax()).reshape(1)
        d = GemmDesc(B=B, M=M, K=K, T=T, ldt=ldt, trans_a=trans_a, k_split=k_split, m_split=m_split, pro_mode=pro_mode,
                     epi_flags=epi_flags, accumulate=accumulate, arith=arith, eps=eps,
                     count=float(count),
                     A=_ptr(A, _f32), A2=_ptr(A2, _f32), X=_ptr(X, _f32), X2=_ptr(X2, _f32), Y=_ptr(Y, _f32), Y2=_ptr(Y2, _f32),
                     bias=_ptr(bias, _f32), pro_alpha=_ptr(pro_alpha, _f32), pro_stats=_ptr(pro_stats, _f64),
                     pro_gamma=_ptr(pro_gamma, _f32), pro_beta=_ptr(pro_beta, _f32), pro_aux=_ptr(pro_aux, _f32),
                     pro_bsum=_ptr(pro_bsum, _f32), pro_bacc=_ptr(pro_bacc, _f64), pro_store=_ptr(pro_store, _f32), pro_dalpha=_ptr(pro_dalpha, _f64),
                     epi_alpha=_ptr(epi_alpha, _f32), epi_stats=_ptr(epi_stats, _f64), epi_res=_ptr(epi_res, _f32),
                     epi_aux=_ptr(epi_aux, _f32), epi_dalpha=_ptr(epi_dalpha, _f64), epi_rowpart=_ptr(epi_rowpart, _f32),
                     a_amax=_ptr(a_amax, _f32), A_pk=_ptr(A_pk.data, _f32) if A_pk is not None else None,
                     a_rscale=_ptr(A_pk.rscale, _f32) if A_pk is not None else None)
        _check(load().sep_pw_gemm(ctypes.byref(d), _stream()), "sep_pw_gemm")

    def pack_weights(self, specs):
        """specs: list of (W, rows, cols, trans) with W a contiguous fp32 view holding a row-major rows x cols matrix.
        Returns one PackedA per spec: A = W (trans = 0, M x K = rows x cols) or W^T (trans = 1, M x K = cols x rows), split
        once into {hi, lo} fp16 groups with a power-of-two scale per row of A (sep_pack_weights; one launch per 64 specs)."""
        if not specs:
            return []
        dev = specs[0][0].device
        total = sum(r * c for _, r, c, _ in specs)
        rows_total = sum((c if t else r) for _, r, c, t in specs)
        data = torch.empty(total + 8, device=dev, dtype=_f32)
        off0 = (-data.data_ptr() // 4) % 8                       # 32-byte alignment of the first segment; sizes are multiples of 8
        rsc = torch.empty(rows_total, device=dev, dtype=_f32)
        _keep(data, rsc)
        arr = (PackSeg * len(specs))()
        out = []
        off, roff = off0, 0
        for i, (W, r, c, t) in enumerate(specs):
            M, K = (c, r) if t else (r, c)
            if K % 8:
                raise SepKernelsError("pack_weights: contraction length {} is not a multiple of 8".format(K))
            pk = PackedA(data[off:off + M * K], rsc[roff:roff + M], M, K, src=(W, r, c, t))
            arr[i] = PackSeg(W=_ptr(W, _f32), dst=pk.data.data_ptr(), rscale=pk.rscale.data_ptr(), M=M, K=K, trans=int(bool(t)), ldw=c)
            out.append(pk)
            off += M * K
            roff += M
        _check(load().sep_pack_weights(arr, len(specs), _stream()), "sep_pack_weights")
        return out

    def pw_wgrad(self, *, B, M, N, T, ldt, G, X, partial, nsplit, G2=None, g_split=0, Gaux=None, g_mul=0, g_div=1,
                 x_mode=PRO_NONE, x_div=1, eps=1e-12, count=0.0, x_alpha=None, x_stats=None, x_gamma=None, x_beta=None,
                 partial_bias=None, arith=None, G2_pre=None):
        """G2_pre: (planes, exps, sums) of the second G source as sep_split_rows wrote them (split_rows below), or None"""
        d = WgradDesc(B=B, M=M, N=N, T=T, ldt=ldt, g_split=g_split, g_mul=g_mul, g_div=g_div, x_mode=x_mode, x_div=x_div,
                      nsplit=nsplit, arith=gemm_arith() if arith is None else arith, eps=eps, count=float(count), G=_ptr(G, _f32), G2=_ptr(G2, _f32), Gaux=_ptr(Gaux, _f32),
                      X=_ptr(X, _f32), x_alpha=_ptr(x_alpha, _f32), x_stats=_ptr(x_stats, _f64), x_gamma=_ptr(x_gamma, _f32),
                      x_beta=_ptr(x_beta, _f32), partial=_ptr(partial, _f32), partial_bias=_ptr(partial_bias, _f32),
                      G2_pre=_ptr(G2_pre[0], _f32) if G2_pre else None, g2_exps=_ptr(G2_pre[1], torch.int32) if G2_pre else None,
                      g2_sums=_ptr(G2_pre[2], _f32) if G2_pre else None)
        _check(load().sep_pw_wgrad(ctypes.byref(d), _stream()), "sep_pw_wgrad")

    def split_rows(self, x, T, k):
        """x (B, C, ldt) -> (planes (B, C, ldt) holding {32 hi | 32 lo} fp16 per 32 frames, exps (B, C) int32, sums (B * k, C)): sep_split_rows"""
        B, C, ldt = x.shape
        planes = torch.empty_like(x)
        exps = torch.empty(B, C, device=x.device, dtype=torch.int32)
        sums = torch.empty(B * k, C, device=x.device, dtype=x.dtype)
        _check(load().sep_split_rows(_ptr(x, _f32), _ptr(planes, _f32), _ptr(exps, torch.int32), _ptr(sums, _f32), B, C, T, ldt, k, _stream()), "sep_split_rows")
        return planes, exps, sums

    def pw_wgrad_batch(self, calls):
        """calls: list (<= 8) of pw_wgrad keyword dicts that agree in everything but G, G2, X, partial, partial_bias: one launch"""
        arr = (WgradDesc * len(calls))()
        for k, c in enumerate(calls):
            c = dict(c)
            g = lambda name, default=None: c.get(name, default)
            arr[k] = WgradDesc(B=c["B"], M=c["M"], N=c["N"], T=c["T"], ldt=c["ldt"], g_split=g("g_split", 0), g_mul=g("g_mul", 0), g_div=g("g_div", 1),
                               x_mode=g("x_mode", PRO_NONE), x_div=g("x_div", 1), nsplit=c["nsplit"], arith=gemm_arith() if g("arith") is None else c["arith"],
                               eps=g("eps", 1e-12), count=float(g("count", 0.0)), G=_ptr(c["G"], _f32), G2=_ptr(g("G2"), _f32), Gaux=_ptr(g("Gaux"), _f32),
                               X=_ptr(c["X"], _f32), x_alpha=_ptr(g("x_alpha"), _f32), x_stats=_ptr(g("x_stats"), _f64), x_gamma=_ptr(g("x_gamma"), _f32),
                               x_beta=_ptr(g("x_beta"), _f32), partial=_ptr(c["partial"], _f32), partial_bias=_ptr(g("partial_bias"), _f32))
        _check(load().sep_pw_wgrad_batch(arr, len(calls), _stream()), "sep_pw_wgrad_batch")

    def reduce_slabs(self, segs):
        """segs: list of (src, src_offset_elems, dst, n, nslab, stride, accumulate, scale)"""
        lib = load()
        for i in range(0, len(segs), 64):
            chunk = segs[i:i + 64]
            arr = (ReduceSeg * len(chunk))()
            for k, (src, off, dst, n, nslab, stride, acc, scale) in enumerate(chunk):
                arr[k] = ReduceSeg(src=_ptr(src, _f32) + 4 * off, dst=_ptr(dst, _f32), n=n, nslab=nslab, stride=stride,
                                   accumulate=acc, scale=scale)
            _check(lib.sep_reduce_slabs(arr, len(chunk), _stream()), "sep_reduce_slabs")

    def f64_to_f32(self, src, dst, n, accumulate=0):
        _check(load().sep_f64_to_f32(_ptr(src, _f64), _ptr(dst, _f32), n, accumulate, _stream()), "sep_f64_to_f32")

    def encoder_fwd(self, x, E, w, stats, B, Cin, Tin, N, L, S, F, ldt, pad_left, relu):
        _check(load().sep_encoder_fwd(_ptr(x, _f32), _ptr(E, _f32), _ptr(w, _f32), _ptr(stats, _f64), B, Cin, Tin, N, L, S, F,
                                      ldt, pad_left, int(relu), _stream()), "sep_encoder_fwd")

    def unfold(self, x, frames, Bp, C, Tin, L, S, F, ldt, pad_left):
        _check(load().sep_unfold(_ptr(x, _f32), _ptr(frames, _f32), Bp, C, Tin, L, S, F, ldt, pad_left, _stream()), "sep_unfold")

    def dwconv_fwd(self, a, stats1, gamma1, beta1, alpha1, wd, bd, alpha2, z, stats2, B, C, T, ldt, dilation, eps):
        _check(load().sep_dwconv_fwd(_ptr(a, _f32), _ptr(stats1, _f64), _ptr(gamma1, _f32), _ptr(beta1, _f32), _ptr(alpha1, _f32),
                                     _ptr(wd, _f32), _ptr(bd, _f32), _ptr(alpha2, _f32), _ptr(z, _f32), _ptr(stats2, _f64),
                                     B, C, T, ldt, dilation, eps, _stream()), "sep_dwconv_fwd")

    def dwconv_bwd(self, dv2, z, a, stats1, gamma1, beta1, alpha1, stats2, gamma2, alpha2, bsum2, wd, bd, dv1, rowpart, bacc1, arrive1, bsum1,
                   B, C, T, ldt, dilation, eps):
        _check(load().sep_dwconv_bwd(_ptr(dv2, _f32), _ptr(z, _f32), _ptr(a, _f32), _ptr(stats1, _f64), _ptr(gamma1, _f32),
                                     _ptr(beta1, _f32), _ptr(alpha1, _f32), _ptr(stats2, _f64), _ptr(gamma2, _f32),
                                     _ptr(alpha2, _f32), _ptr(bsum2, _f32), _ptr(wd, _f32), _ptr(bd, _f32), _ptr(dv1, _f32), _ptr(rowpart, _f32), _ptr(bacc1, _f64),
                                     _ptr(arrive1, torch.int32), _ptr(bsum1, _f32),
                                     B, C, T, ldt, dilation, eps, _stream()), "sep_dwconv_bwd")

    def gln_bwd_finalize(self, rowpart, ntile, nq, stats, gamma, count, eps, bsum, pbeta, pgamma, pextra, B, C):
        _check(load().sep_gln_bwd_finalize(_ptr(rowpart, _f32), ntile, nq, _ptr(stats, _f64), _ptr(gamma, _f32), float(count), eps,
                                           _ptr(bsum, _f32), _ptr(pbeta, _f32), _ptr(pgamma, _f32), _ptr(pextra, _f32), B, C,
                                           _stream()), "sep_gln_bwd_finalize")

    def gln_bwd_finalize_batch(self, segs):
        """segs: list of (rowpart, ntile, nq, stats, gamma, count, eps, bsum, pbeta, pgamma, pextra, B, C) -- sep_gln_bwd_finalize's arguments"""
        lib = load()
        for i in range(0, len(segs), 64):
            chunk = segs[i:i + 64]
            arr = (FinalizeSeg * len(chunk))()
            for k, (rowpart, ntile, nq, stats, gamma, count, eps, bsum, pbeta, pgamma, pextra, B, C) in enumerate(chunk):
                arr[k] = FinalizeSeg(rowpart=_ptr(rowpart, _f32), stats=_ptr(stats, _f64), gamma=_ptr(gamma, _f32), bsum=_ptr(bsum, _f32),
                                     pbeta=_ptr(pbeta, _f32), pgamma=_ptr(pgamma, _f32), pextra=_ptr(pextra, _f32), count=float(count), eps=eps,
                                     ntile=ntile, nq=nq, B=B, C=C)
            _check(lib.sep_gln_bwd_finalize_batch(arr, len(chunk), _stream()), "sep_gln_bwd_finalize_batch")

    def gln_bwd_from_wgrad(self, part, part_bias, W, stats, gamma, beta, count, eps, dW_b, pbeta, pgamma, bacc, arrive, bsum, B, M, N,
                           slabs_per_sample, accumulate=0, products=1):
        _check(load().sep_gln_bwd_from_wgrad(_ptr(part, _f32), _ptr(part_bias, _f32), _ptr(W, _f32), _ptr(stats, _f64), _ptr(gamma, _f32),
                                             _ptr(beta, _f32), float(count), eps, _ptr(dW_b, _f32), _ptr(pbeta, _f32), _ptr(pgamma, _f32),
                                             _ptr(bacc, _f64), _ptr(arrive, torch.int32), _ptr(bsum, _f32), B, M, N, slabs_per_sample, int(accumulate),
                                             int(products), _stream()), "sep_gln_bwd_from_wgrad")

    def head_bwd(self, dvw, w, dwm, stats0, gamma0, bsum0, B, C, T, ldt, count, eps, relu):
        _check(load().sep_head_bwd(_ptr(dvw, _f32), _ptr(w, _f32), _ptr(dwm, _f32), _ptr(stats0, _f64), _ptr(gamma0, _f32),
                                   _ptr(bsum0, _f32), B, C, T, ldt, float(count), eps, int(relu), _stream()), "sep_head_bwd")

    def decoder_fwd(self, w, m, D, est, latent, B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left):
        _check(load().sep_decoder_fwd(_ptr(w, _f32), _ptr(m, _f32), _ptr(D, _f32), _ptr(est, _f32), _ptr(latent, _f32), B, n_src,
                                      N, Cout, L, S, F, ldt, Tout, pad_left, _stream()), "sep_decoder_fwd")

    def decoder_bwd(self, d_est, w, m, D, dpre, dwm, B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left, raw_mask=0):
        _check(load().sep_decoder_bwd(_ptr(d_est, _f32), _ptr(w, _f32), _ptr(m, _f32), _ptr(D, _f32), _ptr(dpre, _f32),
                                      _ptr(dwm, _f32), B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left, int(raw_mask), _stream()),
               "sep_decoder_bwd")

    def softmax_ch_fwd(self, y, B, C, T, ldt):
        _check(load().sep_softmax_ch_fwd(_ptr(y, _f32), B, C, T, ldt, _stream()), "sep_softmax_ch_fwd")

    def softmax_ch_bwd(self, y, g, B, C, T, ldt):
        _check(load().sep_softmax_ch_bwd(_ptr(y, _f32), _ptr(g, _f32), B, C, T, ldt, _stream()), "sep_softmax_ch_bwd")

    def cln_ws_bytes(self, B, C, T, ldt):
        return int(load().sep_cln_ws_bytes(B, C, T, ldt))

    def cln_fwd(self, x, gamma, beta, y, mean, rstd, ws, B, C, T, ldt, eps, alpha=None):
        _check(load().sep_cln_fwd(_ptr(x, _f32), _ptr(gamma, _f32), _ptr(beta, _f32), _ptr(y, _f32), _ptr(mean, _f32), _ptr(rstd, _f32),
                                  _ptr(ws, _f64), B, C, T, ldt, eps, _ptr(alpha, _f32), _stream()), "sep_cln_fwd")

    def cln_bwd(self, dy, x, gamma, mean, rstd, dx, dgamma_part, dbeta_part, ws, B, C, T, ldt, eps, alpha=None, dalpha_part=None):
        _check(load().sep_cln_bwd(_ptr(dy, _f32), _ptr(x, _f32), _ptr(gamma, _f32), _ptr(mean, _f32), _ptr(rstd, _f32), _ptr(dx, _f32),
                                  _ptr(dgamma_part, _f32), _ptr(dbeta_part, _f32), _ptr(ws, _f64), B, C, T, ldt, eps, _ptr(alpha, _f32),
                                  _ptr(dalpha_part, _f32), _stream()), "sep_cln_bwd")

    def attn_fwd(self, qkv, o, lse, N, L, H, D, scale, p_drop=0.0, seed=0):
        _check(load().sep_attn_fwd(_ptr(qkv, _f32), _ptr(o, _f32), _ptr(lse, _f32), N, L, H, D, scale, p_drop, seed, _stream()), "sep_attn_fwd")

    def attn_bwd(self, qkv, o, dout, lse, delta, dqkv, N, L, H, D, scale, p_drop=0.0, seed=0):
        _check(load().sep_attn_bwd(_ptr(qkv, _f32), _ptr(o, _f32), _ptr(dout, _f32), _ptr(lse, _f32), _ptr(delta, _f32), _ptr(dqkv, _f32), N, L, H, D,
                                   scale, p_drop, seed, _stream()), "sep_attn_bwd")

    def rownorm_parts(self, rows, C):
        return int(load().sep_rownorm_parts(rows, C))

    def rownorm_fwd(self, x, res, gamma, beta, s, y, stat, rows, C, eps, p_drop=0.0, seed=0):
        _check(load().sep_rownorm_fwd(_ptr(x, _f32), _ptr(res, _f32), _ptr(gamma, _f32), _ptr(beta, _f32), _ptr(s, _f32), _ptr(y, _f32), _ptr(stat, _f32),
                                      rows, C, eps, p_drop, seed, _stream()), "sep_rownorm_fwd")

    def rownorm_bwd(self, dy, s, gamma, stat, ds, dres, part, rows, C, p_drop=0.0, seed=0):
        _check(load().sep_rownorm_bwd(_ptr(dy, _f32), _ptr(s, _f32), _ptr(gamma, _f32), _ptr(stat, _f32), _ptr(ds, _f32), _ptr(dres, _f32), _ptr(part, _f32),
                                      rows, C, p_drop, seed, _stream()), "sep_rownorm_bwd")

    def relu_drop_fwd(self, h, a, n, p_drop=0.0, seed=0):
        _check(load().sep_relu_drop_fwd(_ptr(h, _f32), _ptr(a, _f32), n, p_drop, seed, _stream()), "sep_relu_drop_fwd")

    def relu_drop_bwd(self, dy, a, dh, n, p_drop=0.0):
        _check(load().sep_relu_drop_bwd(_ptr(dy, _f32), _ptr(a, _f32), _ptr(dh, _f32), n, p_drop, _stream()), "sep_relu_drop_bwd")

    def gln_tokens_ws_bytes(self, nseq, L, C):
        return int(load().sep_gln_tokens_ws_bytes(nseq, L, C))

    def gln_tokens_fwd(self, x, gamma, beta, y, stats, nseq, L, C, eps, ws=None):
        _check(load().sep_gln_tokens_fwd(_ptr(x, _f32), _ptr(gamma, _f32), _ptr(beta, _f32), _ptr(y, _f32), _ptr(stats, _f32), _ptr(ws, _f64), nseq, L, C,
                                         eps, _stream()), "sep_gln_tokens_fwd")

    def gln_tokens_bwd(self, dy, x, gamma, stats, dx, part, nseq, L, C, ws=None):
        _check(load().sep_gln_tokens_bwd(_ptr(dy, _f32), _ptr(x, _f32), _ptr(gamma, _f32), _ptr(stats, _f32), _ptr(dx, _f32), _ptr(part, _f32),
                                         _ptr(ws, _f64), nseq, L, C, _stream()), "sep_gln_tokens_bwd")

    def gln_stats(self, x, stats, B, C, T, ldt):
        _check(load().sep_gln_stats(_ptr(x, _f32), _ptr(stats, _f64), B, C, T, ldt, _stream()), "sep_gln_stats")

    def gln_apply(self, x, stats, gamma, beta, y, B, C, T, ldt, count, eps):
        _check(load().sep_gln_apply(_ptr(x, _f32), _ptr(stats, _f64), _ptr(gamma, _f32), _ptr(beta, _f32), _ptr(y, _f32), B, C, T,
                                    ldt, float(count), eps, _stream()), "sep_gln_apply")

    def gln_bwd_rowsums(self, dy, x, rowpart, B, C, T, ldt):
        _check(load().sep_gln_bwd_rowsums(_ptr(dy, _f32), _ptr(x, _f32), _ptr(rowpart, _f32), B, C, T, ldt, _stream()),
               "sep_gln_bwd_rowsums")

    def gln_bwd_apply(self, dy, x, stats, gamma, bsum, dx, B, C, T, ldt, count, eps):
        _check(load().sep_gln_bwd_apply(_ptr(dy, _f32), _ptr(x, _f32), _ptr(stats, _f64), _ptr(gamma, _f32), _ptr(bsum, _f32),
                                        _ptr(dx, _f32), B, C, T, ldt, float(count), eps, _stream()), "sep_gln_bwd_apply")

    def repack(self, src, ld_src, dst, ld_dst, rows, T):
        _check(load().sep_repack(_ptr(src, _f32), ld_src, _ptr(dst, _f32), ld_dst, rows, T, _stream()), "sep_repack")

    def depthwise_fwd(self, x, w, bias, y, B, C, Tin, Tout, Kw, stride, pad, dil):
        _check(load().sep_depthwise_fwd(_ptr(x, _f32), _ptr(w, _f32), _ptr(bias, _f32), _ptr(y, _f32), B, C, Tin, Tout, Kw, stride,
                                        pad, dil, _stream()), "sep_depthwise_fwd")

    def depthwise_bwd_input(self, dy, w, dx, B, C, Tin, Tout, Kw, stride, pad, dil):
        _check(load().sep_depthwise_bwd_input(_ptr(dy, _f32), _ptr(w, _f32), _ptr(dx, _f32), B, C, Tin, Tout, Kw, stride, pad, dil,
                                              _stream()), "sep_depthwise_bwd_input")

    def depthwise_bwd_weight(self, dy, x, partial, B, C, Tin, Tout, Kw, stride, pad, dil):
        _check(load().sep_depthwise_bwd_weight(_ptr(dy, _f32), _ptr(x, _f32), _ptr(partial, _f32), B, C, Tin, Tout, Kw, stride, pad,
                                               dil, _stream()), "sep_depthwise_bwd_weight")

    def segment(self, x, out, rows, T, ldt, S, chunk, hop, pad_left):
        _check(load().sep_segment(_ptr(x, _f32), _ptr(out, _f32), rows, T, ldt, S, chunk, hop, pad_left, _stream()), "sep_segment")

    def overlap_add(self, y, out, rows, T, ldt, S, chunk, hop, pad_left):
        _check(load().sep_overlap_add(_ptr(y, _f32), _ptr(out, _f32), rows, T, ldt, S, chunk, hop, pad_left, _stream()), "sep_overlap_add")

    def sisdr_dots(self, est, tgt, dots, tt, xx, B, n, T, all_pairs):
        _check(load().sep_sisdr_dots(_ptr(est, _f32), _ptr(tgt, _f32), _ptr(dots, _f64), _ptr(tt, _f64), _ptr(xx, _f64), B, n, T,
                                     int(all_pairs), _stream()), "sep_sisdr_dots")

    def sisdr_from_dots(self, dots, tt, xx, out, B, n, all_pairs, eps):
        _check(load().sep_sisdr_from_dots(_ptr(dots, _f64), _ptr(tt, _f64), _ptr(xx, _f64), _ptr(out, _f32), B, n, int(all_pairs),
                                          eps, _stream()), "sep_sisdr_from_dots")

    def sisdr_bwd(self, est, tgt, dots, tt, xx, gw, d_est, B, n, T, all_pairs, eps):
        _check(load().sep_sisdr_bwd(_ptr(est, _f32), _ptr(tgt, _f32), _ptr(dots, _f64), _ptr(tt, _f64), _ptr(xx, _f64),
                                    _ptr(gw, _f32), _ptr(d_est, _f32), B, n, T, int(all_pairs), eps, _stream()), "sep_sisdr_bwd")

    def pit_search(self, val, perms, P, n, B, maximize, use_mean, best_val, best_idx):
        _check(load().sep_pit_search(_ptr(val, _f32), _ptr(perms, torch.int32), P, n, B, int(maximize), int(use_mean),
                                     _ptr(best_val, _f32), _ptr(best_idx, torch.int64), _stream()), "sep_pit_search")

    def sinkhorn_fwd(self, C, zwork, loss, P, B, n, coldness, iters):
        _check(load().sep_sinkhorn_fwd(_ptr(C, _f32), _ptr(zwork, _f64), _ptr(loss, _f32), _ptr(P, _f32), B, n, coldness, iters,
                                       _stream()), "sep_sinkhorn_fwd")

    def sinkhorn_bwd(self, C, zwork, dloss, dC, B, n, coldness, iters):
        _check(load().sep_sinkhorn_bwd(_ptr(C, _f32), _ptr(zwork, _f64), _ptr(dloss, _f32), _ptr(dC, _f32), B, n, coldness, iters,
                                       _stream()), "sep_sinkhorn_bwd")

    def rowdiff_sums(self, x, t, sums, rows, T):
        _check(load().sep_rowdiff_sums(_ptr(x, _f32), _ptr(t, _f32), _ptr(sums, _f64), rows, T, _stream()), "sep_rowdiff_sums")

    def rowdiff_bwd(self, x, t, c_abs, c_sq, dx, rows, T):
        _check(load().sep_rowdiff_bwd(_ptr(x, _f32), _ptr(t, _f32), _ptr(c_abs, _f32), _ptr(c_sq, _f32), _ptr(dx, _f32), rows, T,
                                      _stream()), "sep_rowdiff_bwd")

    def sqnorm(self, g, out, n):
        _check(load().sep_sqnorm(_ptr(g, _f32), _ptr(out, _f64), n, _stream()), "sep_sqnorm")

    def lstm_fwd(self, xg, w_hh, h_out, gates, cstate, nseq, L, H, reverse):
        _check(load().sep_lstm_fwd(_ptr(xg, _f32), _ptr(w_hh, _f32), _ptr(h_out, _f32), _ptr(gates, _f32), _ptr(cstate, _f32),
                                   nseq, L, H, int(reverse), _stream()), "sep_lstm_fwd")

    def lstm_bwd(self, dh_out, gates, cstate, w_hh, dxg, nseq, L, H, reverse):
        _check(load().sep_lstm_bwd(_ptr(dh_out, _f32), _ptr(gates, _f32), _ptr(cstate, _f32), _ptr(w_hh, _f32), _ptr(dxg, _f32),
                                   nseq, L, H, int(reverse), _stream()), "sep_lstm_bwd")

    def chunk_to_tokens(self, x, y, B, F, S, K, inter):
        _check(load().sep_chunk_to_tokens(_ptr(x, _f32), _ptr(y, _f32), B, F, S, K, int(inter), _stream()), "sep_chunk_to_tokens")

    def tokens_to_chunk(self, y, x, B, F, S, K, inter):
        _check(load().sep_tokens_to_chunk(_ptr(y, _f32), _ptr(x, _f32), B, F, S, K, int(inter), _stream()), "sep_tokens_to_chunk")

    def linear_fwd(self, x, w, bias, bias2, y, ntok, K, N):
        _check(load().sep_linear_fwd(_ptr(x, _f32), _ptr(w, _f32), _ptr(bias, _f32), _ptr(bias2, _f32), _ptr(y, _f32), ntok, K, N, _stream()), "sep_linear_fwd")

    def linear_bwd_input(self, dy, w, dx, ntok, K, N, accumulate):
        _check(load().sep_linear_bwd_input(_ptr(dy, _f32), _ptr(w, _f32), _ptr(dx, _f32), ntok, K, N, int(accumulate), _stream()), "sep_linear_bwd_input")

    def linear_bwd_weight(self, dy, x, ldx, partial, partial_bias, ntok, K, N, L, shift, nslab):
        # x may be a column slice of a wider row-major matrix (rows ldx floats apart): one direction's half of an interleaved bi-LSTM output
        if not (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) == ldx):
            raise SepKernelsError("linear_bwd_weight: x must be a matrix with unit column stride and row stride ldx")
        _ptr(x[:1], _f32)                                     # device / dtype checks on a (contiguous) row of it
        _check(load().sep_linear_bwd_weight(_ptr(dy, _f32), x.data_ptr(), ldx, _ptr(partial, _f32), _ptr(partial_bias, _f32), ntok, K, N, L, shift,
                                            nslab, _stream()), "sep_linear_bwd_weight")

    def memset(self, t, value=0):
        _check(load().sep_memset(_ptr(t), int(value), t.numel() * t.element_size(), _stream()), "sep_memset")

    def zeros(self, *shape, device, dtype):
        """torch.zeros, except while a Sequence is being recorded: the clearing is then a launch of the sequence (sep_memset)"""
        if _recording is None:
            return torch.zeros(*shape, device=device, dtype=dtype)
        t = torch.empty(*shape, device=device, dtype=dtype)
        self.memset(t, 0)
        return t

    def absmax(self, x, out, n):
        _check(load().sep_absmax(_ptr(x, _f32), n, _ptr(out, _f32), _stream()), "sep_absmax")

    def pit_finish(self, best_val, best_idx, perms, P, n, B, sign, scale, loss, gw, pattern):
        _check(load().sep_pit_finish(_ptr(best_val, _f32), _ptr(best_idx, torch.int64), _ptr(perms, torch.int32), P, n, B, sign, scale,
                                     _ptr(loss, _f32), _ptr(gw, _f32), _ptr(pattern, torch.int64), _stream()), "sep_pit_finish")

    def axpby(self, x, a, y, b, out, n):
        _check(load().sep_axpby(_ptr(x, _f32), a, _ptr(y, _f32), b, _ptr(out, _f32), n, _stream()), "sep_axpby")

    def adam_step(self, p, g, m, v, sqnorm, n, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, step):
        _check(load().sep_adam_step(_ptr(p, _f32), _ptr(g, _f32), _ptr(m, _f32), _ptr(v, _f32), _ptr(sqnorm, _f64), n, lr, beta1,
                                    beta2, eps, weight_decay, max_norm, grad_scale, step, _stream()), "sep_adam_step")


    def adam_step_dev(self, p, g, m, v, sqnorm, n, lr_dev, step_dev, beta1, beta2, eps, weight_decay, max_norm, grad_scale):
        _check(load().sep_adam_step_dev(_ptr(p, _f32), _ptr(g, _f32), _ptr(m, _f32), _ptr(v, _f32), _ptr(sqnorm, _f64), n, _ptr(lr_dev, _f32),
                                        _ptr(step_dev, torch.int32), beta1, beta2, eps, weight_decay, max_norm, grad_scale, _stream()), "sep_adam_step_dev")


_backend = HipBackend()


def backend():
    return _backend


def _set_backend_for_tests(b):
    """Test hook (tests/emulator.py): swap the kernel facade for the CPU emulator so the host-side orchestration
    can be checked without a GPU.  Never called by product code."""
    global _backend
    old = _backend
    _backend = b
    return old
