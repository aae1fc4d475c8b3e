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
    assert lib.sep_version() == sepkernels.ABI_VERSION == 23
    header = open(HEADER).read()
    assert "#define SEP_ABI_VERSION 23" in header and "#define SEP_STATS_SLOTS 16" in header
    assert sepkernels.STATS_SLOTS == 16


def test_descriptor_layouts_match_the_header():
    # both: 12 int32 + float (+4 pad), double at offset 56, then pointers: the layouts the header's field order implies on LP64
    assert sepkernels.GemmDesc.arith.offset == 44 and sepkernels.GemmDesc.eps.offset == 48
    assert sepkernels.GemmDesc.count.offset == 56 and sepkernels.GemmDesc.A.offset == 64
    assert ctypes.sizeof(sepkernels.GemmDesc) == 64 + 25 * 8
    assert sepkernels.GemmDesc.A_pk.offset == 64 + 23 * 8 and ctypes.sizeof(sepkernels.PackSeg) == 40
    assert sepkernels.WgradDesc.arith.offset == 44 and sepkernels.WgradDesc.count.offset == 56 and sepkernels.WgradDesc.G.offset == 64
    assert ctypes.sizeof(sepkernels.WgradDesc) == 64 + 13 * 8 and sepkernels.WgradDesc.G2_pre.offset == 64 + 10 * 8      # (ABI 23: G2_pre, g2_exps, g2_sums)
    assert ctypes.sizeof(sepkernels.ReduceSeg) == 40


def test_sequence_table_covers_every_entry_point_that_takes_a_stream():
    """ABI 23: sep_run_sequence replays recorded calls of the library's own entry points.  Every declared entry point whose last parameter
    is the stream has an id, its argument count is the header's, the op record has the C layout, and a bad op comes back as an error
    naming it (no launch happens here: the failing ops fail their own argument checks before any HIP call)."""
    lib = sepkernels.load()
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    decls = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(?:int|size_t|const char\s*\*)\s+(sep_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S))
    with_stream = sorted(n for n, a in decls.items() if a.strip().endswith("sep_stream_t stream") and n != "sep_run_sequence")
    assert lib.sep_seq_count() == len(with_stream)
    ids = set()
    for n in with_stream:
        k = lib.sep_seq_lookup(n.encode())
        assert k >= 0, n
        ids.add(k)
        assert lib.sep_seq_name(k).decode() == n
        assert lib.sep_seq_nargs(k) == len(decls[n].split(",")) - 1 == len(sepkernels.SIGNATURES[n]) - 1, n
        assert lib.sep_seq_nargs(k) <= sepkernels.SEQ_MAX_ARGS
    assert ids == set(range(len(with_stream)))
    for n in ("sep_version", "sep_last_error", "sep_cln_ws_bytes", "sep_run_sequence", "nonsense"):
        assert lib.sep_seq_lookup(n.encode()) == -1
    assert ctypes.sizeof(sepkernels.SeqArg) == 8 and sepkernels.SeqOp.args.offset == 8
    assert ctypes.sizeof(sepkernels.SeqOp) == 8 + 8 * sepkernels.SEQ_MAX_ARGS
    assert "#define SEP_SEQ_MAX_ARGS {}".format(sepkernels.SEQ_MAX_ARGS) in open(HEADER).read()
    # an empty list is fine; a bad id, a wrong argument count and an op that fails its own checks are errors that name the op
    assert lib.sep_run_sequence(None, 0, None) == 0
    ops = (sepkernels.SeqOp * 2)()
    ops[0].fn, ops[0].nargs = 999, 0
    assert lib.sep_run_sequence(ops, 1, None) < 0 and b"op 0 names entry point 999" in lib.sep_last_error()
    k = lib.sep_seq_lookup(b"sep_sqnorm")
    ops[0].fn, ops[0].nargs = k, 1
    assert lib.sep_run_sequence(ops, 1, None) < 0 and b"sep_sqnorm" in lib.sep_last_error() and b"carries 1 arguments" in lib.sep_last_error()
    ops[0].fn, ops[0].nargs = lib.sep_seq_lookup(b"sep_memset"), 3          # memset of zero bytes: succeeds without touching the device
    ops[1].fn, ops[1].nargs = k, 3                                          # sep_sqnorm(NULL, NULL, 0): its own argument check fails
    assert lib.sep_run_sequence(ops, 2, None) < 0
    msg = lib.sep_last_error()
    assert b"op 1 of 2 (sep_sqnorm)" in msg and b"bad arguments" in msg, msg


def test_recording_proxy_marshals_arguments_by_signature():
    """sepkernels.Sequence.add: pointers / host descriptors / integers / floats land in the right member of the op's argument union, host
    descriptors are kept alive, entry points without a stream are passed through unrecorded."""
    lib = sepkernels.load()
    seq = sepkernels.Sequence()
    d = sepkernels.GemmDesc(B=1, M=128, K=100, T=10, ldt=128)
    with sepkernels.recording(seq):
        rec = sepkernels.load()
        assert rec.sep_version() == 23                                       # not a launch: passed through
        assert rec.sep_pw_gemm(ctypes.byref(d), None) < 0                    # a failing call is not recorded
        assert rec.sep_memset(None, 0, 0, None) == 0
        with pytest.raises(sepkernels.SepKernelsError):
            with sepkernels.recording(sepkernels.Sequence()):
                pass
    assert not sepkernels.is_recording() and seq.names() == ["sep_memset"]
    seq.add(lib, "sep_pw_gemm", (ctypes.byref(d), None))
    seq.add(lib, "sep_adam_step_dev", (1, 2, 3, 4, 5, 77, 6, 7, 0.9, 0.999, 1e-8, 0.0, 5.0, 1.0, None))
    arr = seq._array()
    assert arr[1].fn == lib.sep_seq_lookup(b"sep_pw_gemm") and arr[1].nargs == 1 and arr[1].args[0].p == ctypes.addressof(d) and seq.keep[-1] is d
    assert arr[2].nargs == 14 and arr[2].args[5].i == 77 and arr[2].args[4].p == 5 and abs(arr[2].args[9].f - 0.999) < 1e-12


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
