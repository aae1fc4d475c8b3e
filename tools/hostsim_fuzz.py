"""Development tool: random small shapes through the kernel sources on the host simulation (tools/hostsim.py), against the restatement of
tests/emulator.py -- ragged frame counts, widths off every tile size, each arithmetic.  The fixed cases of tests/test_gpu_kernels.py
cover the shapes the models use; this looks for the ones nobody thought of.  A failure here is a failure on the device too (same source).

    python tools/hostsim_fuzz.py [seconds] [seed]
"""
import os
import random
import sys
import tempfile
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src"), ROOT]
import hostsim                      # noqa: E402
import sepkernels                   # noqa: E402
import test_gpu_kernels as GK       # noqa: E402
from sepkernels import PRO_PRELU, EPI_SIGMOID      # noqa: E402


def up(a, b):
    return (a + b - 1) // b * b


def case_gemm(R):
    arith = R.choice(["f32", "f16x3", "bf16x6", "f16x3-packed"])
    B, T = R.randint(1, 3), R.randint(1, 300)
    M, K = 4 * R.randint(1, 70), 16 * R.randint(1, 9)
    if arith.endswith("packed"):                       # weights split beforehand by sep_pack_weights: 32-row blocks; long contractions and
        M = 32 * R.randint(1, 10)                      # tall outputs reach the producer / consumer kernel
        K = R.choice([16, 64, 128, 256, 512, 640])
    trans = R.random() < 0.4
    ldt = up(T, 128)
    X = GK.padded(B, K, T, ldt)
    A = GK.rnd(K, M, scale=K ** -0.5) if trans else GK.rnd(M, K, scale=K ** -0.5)
    kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=GK.nan(B, M, ldt), trans_a=int(trans))
    if R.random() < 0.6:
        kw["bias"] = GK.rnd(M)
    if R.random() < 0.3:
        kw.update(pro_mode=PRO_PRELU, pro_alpha=torch.tensor([R.uniform(-0.5, 1.5)]))
    if R.random() < 0.2 and not trans:
        kw["epi_flags"] = EPI_SIGMOID
    prev = sepkernels.set_gemm_arith(arith.split("-")[0])
    GK.PACKED[0] = arith.endswith("packed")
    try:
        GK.both("pw_gemm", [], kw)
    finally:
        GK.PACKED[0] = False
        sepkernels.set_gemm_arith(prev)
    return "pw_gemm {} B={} M={} K={} T={} trans={} {}".format(arith, B, M, K, T, trans, sorted(k for k in kw if k in ("bias", "pro_mode", "epi_flags")))


def walk(n_terms):
    """start value of a slope-gradient accumulator: the kernels add onto it, and a sum of n signed terms is only known to
    about sqrt(n) of a term -- a zero start would gate the comparison on whatever the terms happen to cancel to"""
    return torch.full((1,), float(n_terms) ** 0.5, dtype=torch.float64)


def case_gemm_forms(R):
    """the prologue / epilogue forms of the model's products at random sizes: gLN (+PReLU) prologue with statistics or residual epilogue
    (optionally two outputs + accumulation), input gradients from two sources with row sums, PReLU-derivative epilogue, gLN-backward
    prologue -- fp32 weights or packed ones"""
    from sepkernels import EPI_PRELU_BWD, EPI_RESIDUAL, EPI_ROWSUMS, EPI_ROWSUMS_PRELU, EPI_STATS_PRELU, PRO_GLN, PRO_GLN_BWD, PRO_GLN_PRELU
    arith = R.choice(["f32", "f16x3", "bf16x6", "f16x3-packed"])
    packed = arith.endswith("packed")
    B, T = R.randint(1, 2), R.randint(1, 400)
    ldt = up(T, 128)
    unit = 32 if packed else 16
    form = R.randint(0, 4)
    al = torch.tensor([R.choice([0.25, -0.3, 1.0, 1.7, 0.0])])
    if form == 0:          # conv1: statistics of PReLU(y) in the epilogue, optionally behind the gLN prologue
        M, K = unit * R.randint(1, 12), 16 * R.randint(1, 8)
        X = GK.padded(B, K, T, ldt) * 2 + 0.3
        X[..., T:] = 0
        kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, A=GK.rnd(M, K, scale=K ** -0.5), X=X, Y=GK.nan(B, M, ldt), bias=GK.rnd(M), epi_flags=EPI_STATS_PRELU,
                  epi_alpha=al, epi_stats=GK.zstats(B), eps=1e-12)
        if R.random() < 0.5:
            kw.update(pro_mode=PRO_GLN, pro_stats=GK.stats_of(X, T), pro_gamma=GK.rnd(K) + 1, pro_beta=GK.rnd(K), count=K * T)
    elif form == 1:        # heads: gLN + PReLU prologue, residual, optionally [out; skip] with accumulation into the skip sum
        K = 16 * R.randint(1, 8)
        z = GK.padded(B, K, T, ldt) * 1.5 + 0.2
        z[..., T:] = 0
        st = GK.stats_of(torch.where(z > 0, z, al * z), T)
        common = dict(B=B, K=K, T=T, ldt=ldt, X=z, pro_mode=PRO_GLN_PRELU, pro_stats=st, pro_gamma=GK.rnd(K) + 1, pro_beta=GK.rnd(K), pro_alpha=al, count=K * T, eps=1e-12)
        if R.random() < 0.5:
            Bn, Sc = 128 * R.randint(1, 2), unit * R.randint(1, 4)
            kw = dict(common, M=Bn + Sc, A=GK.rnd(Bn + Sc, K, scale=K ** -0.5), Y=GK.nan(B, Bn, ldt), Y2=GK.padded(B, Sc, T, ldt), m_split=Bn, bias=GK.rnd(Bn + Sc),
                      accumulate=1, epi_flags=EPI_RESIDUAL, epi_res=GK.padded(B, Bn, T, ldt))
        else:
            M = unit * R.randint(1, 8)
            kw = dict(common, M=M, A=GK.rnd(M, K, scale=K ** -0.5), Y=GK.nan(B, M, ldt), bias=GK.rnd(M))
            if R.random() < 0.5:
                kw.update(epi_flags=EPI_RESIDUAL, epi_res=GK.padded(B, M, T, ldt))
    elif form == 2:        # heads^T: two sources, row sums for the gLN backward
        Bn, Sc, H = 16 * R.randint(1, 8), 16 * R.randint(1, 4), unit * R.randint(1, 8)
        kw = dict(B=B, M=H, K=Bn + Sc, T=T, ldt=ldt, trans_a=1, A=GK.rnd(Bn, H, scale=0.1), A2=GK.rnd(Sc, H, scale=0.1), X=GK.padded(B, Bn, T, ldt), X2=GK.padded(B, Sc, T, ldt),
                  k_split=Bn, Y=GK.nan(B, H, ldt), epi_flags=EPI_ROWSUMS | EPI_ROWSUMS_PRELU, epi_aux=GK.padded(B, H, T, ldt), epi_alpha=al, epi_rowpart=GK.nan(B, H, ldt // 64, 2))
        if R.random() < 0.3:
            kw.update(epi_flags=EPI_ROWSUMS)
    elif form == 3:        # mask^T: PReLU-derivative epilogue with the slope gradient
        M, K = unit * R.randint(1, 6), 16 * R.randint(1, 12)
        kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, trans_a=1, A=GK.rnd(K, M, scale=0.1), X=GK.padded(B, K, T, ldt), Y=GK.nan(B, M, ldt), epi_flags=EPI_PRELU_BWD,
                  epi_aux=GK.padded(B, M, T, ldt), epi_alpha=al, epi_dalpha=walk(B * M * T))
    else:                  # conv1^T: gLN-backward prologue (stores d(pre-activation), accumulates the slope gradient), optional residual
        M, K = unit * R.randint(1, 6), 16 * R.randint(1, 10)
        a = GK.padded(B, K, T, ldt)
        dv = GK.padded(B, K, T, ldt)
        kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, trans_a=1, A=GK.rnd(K, M, scale=0.1), X=dv, Y=GK.nan(B, M, ldt), pro_mode=PRO_GLN_BWD,
                  pro_stats=GK.stats_of(torch.where(a > 0, a, al * a), T), pro_gamma=GK.rnd(K) + 1, pro_alpha=al, pro_aux=a, pro_bsum=GK.rnd(B, 2, scale=0.01), pro_store=dv if M <= 128 else GK.nan(B, K, ldt),     # in place only with one row tile (sepkernels.h)
                  pro_dalpha=walk(B * K * T), count=K * T, eps=1e-12)
        if R.random() < 0.5:
            kw.update(epi_flags=EPI_RESIDUAL, epi_res=GK.padded(B, M, T, ldt))
    what = "gemm form {} {} B={} M={} K={} T={} k_split={} m_split={} flags={} pro={}".format(
        form, arith, B, kw["M"], kw["K"], T, kw.get("k_split", 0), kw.get("m_split", 0), kw.get("epi_flags", 0), kw.get("pro_mode", 0))
    prev = sepkernels.set_gemm_arith(arith.split("-")[0])
    GK.PACKED[0] = packed
    try:
        GK.both("pw_gemm", [], kw)
    except AssertionError as e:
        raise AssertionError("{}: {}".format(what, e)) from None
    finally:
        GK.PACKED[0] = False
        sepkernels.set_gemm_arith(prev)
    return what


def case_wgrad(R):
    arith = R.choice(["f32", "f16x3", "bf16x6"])
    B, T = R.randint(1, 3), R.randint(1, 300)
    M, N = R.choice([4, 16, 24, 32, 64, 96, 128, 160, 256]), R.choice([2, 4, 16, 20, 32, 64, 128, 256])
    ldt = up(T, 128)
    ns = R.randint(1, min(12, B * (ldt // 32)))
    if R.random() < 0.4:       # sample-aligned slabs (what sep_gln_bwd_from_wgrad takes): B * k with k a divisor of ldt / 32
        ns = B * R.choice([q for q in range(1, ldt // 32 + 1) if (ldt // 32) % q == 0])
    part, pb = GK.nan(ns, M, N), GK.nan(ns, M)
    prev = sepkernels.set_gemm_arith(arith)
    try:
        kw = dict(B=B, M=M, N=N, T=T, ldt=ldt, G=GK.padded(B, M, T, ldt), X=GK.padded(B, N, T, ldt), partial=part, partial_bias=pb, nsplit=ns)
        GK._wgrad_both(kw)
    finally:
        sepkernels.set_gemm_arith(prev)
    return "pw_wgrad {} B={} M={} N={} T={} ns={}".format(arith, B, M, N, T, ns)


def case_codec(R):
    B, Cin = R.randint(1, 2), R.randint(1, 2)
    S = R.choice([1, 2, 4, 8, 10])
    L = S * R.choice([1, 2, 4])
    N = R.choice([4, 16, 24, 32, 64])
    Tin = R.randint(L, L + 1500)                        # the TasNet forwards' geometry (conv_tasnet.py:145-149): pad to a whole number of hops
    padding = (S - (Tin - L) % S) % S
    pad_left = padding // 2
    F = (Tin + padding - L) // S + 1
    ldt = up(F, 128)
    relu = R.randint(0, 1)
    x, E = GK.rnd(B, Cin, Tin), GK.rnd(N, Cin, L)
    GK.both("encoder_fwd", [x, E, GK.nan(B, N, ldt), GK.zstats(B), B, Cin, Tin, N, L, S, F, ldt, pad_left, relu])
    n_src = R.randint(1, 4)
    w, m, D = GK.padded(B, N, F, ldt), GK.padded(B, n_src * N, F, ldt), GK.rnd(N, Cin, L)
    Tout = Tin
    GK.both("decoder_fwd", [w, m, D, GK.nan(B, n_src, Cin, Tout), None, B, n_src, N, Cin, L, S, F, ldt, Tout, pad_left])
    GK.both("decoder_bwd", [GK.rnd(B, n_src, Cin, Tout), w, m, D, GK.nan(B, n_src * N, ldt), GK.nan(B, N, ldt), B, n_src, N, Cin, L, S, F, ldt, Tout, pad_left],
            dict(raw_mask=R.randint(0, 1)))
    return "codec B={} Cin={} N={} L={} S={} F={} Tin={} pad_left={} n_src={}".format(B, Cin, N, L, S, F, Tin, pad_left, n_src)


def case_norms(R):
    B, C, T = R.randint(1, 3), R.choice([1, 3, 8, 20, 64]), R.randint(1, 700)
    ldt = up(T, 4)
    x = GK.padded(B, C, T, ldt)
    st = GK.zstats(B)
    GK.both("gln_stats", [x, st, B, C, T, ldt])
    GK.both("gln_apply", [x, GK.stats_of(x, T), GK.rnd(C) + 1, GK.rnd(C), GK.nan(B, C, ldt), B, C, T, ldt, C * float(T), 1e-12])
    if C >= 3:          # (one channel: the first frame has zero variance and 1 / (sigma + eps) = 1e12 either way -- nothing to compare)
        xs = x * 2 + 0.1 * (x != 0)
        ye, me, re_ = GK.nan(B, C, ldt), torch.zeros(B, ldt), torch.zeros(B, ldt)          # ABI 20: the per-frame statistics in rows of ldt
        ys, ms, rs = GK.nan(B, C, ldt), torch.zeros(B, ldt), torch.zeros(B, ldt)
        gamma, beta = GK.rnd(C) + 1, GK.rnd(C)
        GK.EMU.cln_fwd(xs, gamma, beta, ye, me, re_, torch.empty(B, 2, ldt, dtype=torch.float64), B, C, T, ldt, 1e-12)
        GK.HIP.cln_fwd(xs.clone(), gamma.clone(), beta.clone(), ys, ms, rs, torch.empty((GK.HIP.cln_ws_bytes(B, C, T, ldt) + 7) // 8, dtype=torch.float64), B, C, T, ldt, 1e-12)   # the workspace is scratch
        assert torch.isfinite(ys).all() and (ys - ye).abs().max() <= 5e-4 * ye.abs().max() and (ms[:, :T] - me[:, :T]).abs().max() <= 1e-5 * (1 + me.abs().max())
    return "norms B={} C={} T={}".format(B, C, T)


def case_chunks(R):
    B, C = R.randint(1, 2), R.randint(1, 5)
    hop = R.randint(1, 40)
    chunk = hop * R.randint(1, 3) if R.random() < 0.7 else R.randint(hop, 3 * hop)
    T = R.randint(1, 500)
    from sepkernels.functional import segment_geometry
    pad_left, _, S = segment_geometry(T, chunk, hop)
    if S < 1:
        return "chunks skipped"
    ldt = up(T, 4)
    x = GK.padded(B, C, T, ldt)
    GK.both("segment", [x, GK.nan(B, C, S, chunk), B * C, T, ldt, S, chunk, hop, pad_left], tol=0.0)
    GK.both("overlap_add", [GK.rnd(B, C, S, chunk), GK.nan(B, C, ldt), B * C, T, ldt, S, chunk, hop, pad_left], tol=1e-6)
    return "chunks B={} C={} T={} chunk={} hop={} S={}".format(B, C, T, chunk, hop, S)


def case_lstm(R):
    H, nseq, L, rev = R.choice([16, 32]), R.randint(1, 40), R.randint(1, 12), R.randint(0, 1)
    kernel = R.choice(["sixteen", "four"])
    GK.test_lstm_sweeps(H, nseq, L, rev, kernel)
    return "lstm H={} nseq={} L={} reverse={} kernel={}".format(H, nseq, L, rev, kernel)


def case_dense(R):
    """token-major dense layers and the chunk <-> token layout pair (csrc/linear.hip) at random sizes: ragged token tiles, both tile widths,
    slab counts beyond the token count, x as a column slice of a wider matrix, both shifts"""
    pick = R.randint(0, 3)
    if pick == 0:
        ntok, K, N = R.randint(1, 700), 64 * R.randint(1, 4), 64 * R.randint(1, 8)
        GK.test_linear_forward_and_input_gradient(ntok, K, N)
        return "linear fwd / dx ntok={} K={} N={}".format(ntok, K, N)
    if pick == 1:
        nseq, L, K, N = R.randint(1, 12), R.randint(1, 40), 64 * R.randint(1, 4), 64 * R.randint(1, 8)
        shift, nslab = R.choice([-1, 0, 1]), R.randint(1, 30)
        GK.test_linear_weight_gradient(nseq, L, K, N, shift, nslab)
        return "linear dW nseq={} L={} K={} N={} shift={} nslab={}".format(nseq, L, K, N, shift, nslab)
    if pick == 2:
        B, F, S, K = R.randint(1, 3), R.randint(1, 70), R.randint(1, 9), R.randint(1, 80)
        GK.test_chunk_tokens_layout_pair(B, F, S, K)
        return "chunk <-> tokens B={} F={} S={} K={}".format(B, F, S, K)
    kernel = R.choice(["sixteen", "four"])
    GK.test_lstm_sweeps_interleaved_output(kernel)
    return "lstm interleaved " + kernel


def case_test_functions(R):
    """the parametrised kernel tests of tests/test_gpu_kernels.py at random parameters"""
    pick = R.randint(0, 6)
    if pick == 0:
        T, d = R.randint(1, 1500), R.choice([1, 2, 4, 8, 16, 32, 64, 128, 256])
        GK.test_dwconv_fwd_bwd(T, d)
        return "dwconv T={} d={}".format(T, d)
    if pick == 1:
        B, C, T = R.randint(1, 3), R.randint(1, 300), R.randint(1, 400)
        GK.test_softmax_over_channels(B, C, T)
        return "softmax {} {} {}".format(B, C, T)
    if pick == 2:
        n, mx, mean = R.randint(1, 5), R.randint(0, 1), R.randint(0, 1)
        GK.test_pit_search(n, mx, mean)
        return "pit_search {} {} {}".format(n, mx, mean)
    if pick == 3:
        n, it, beta = R.randint(1, 8), R.randint(1, 30), R.choice([0.5, 1.0, 2.0])
        GK.test_sinkhorn(n, it, beta)
        return "sinkhorn {} {} {}".format(n, it, beta)
    if pick == 4:
        rows, T = R.randint(1, 20), R.randint(1, 3000)
        GK.test_rowdiff_sums_and_bwd(rows, T)
        return "rowdiff {} {}".format(rows, T)
    if pick == 5:
        Kw, stride, dil = R.randint(1, 7), R.randint(1, 4), R.randint(1, 4)
        pad = R.randint(0, 6)
        Tin = R.randint(dil * (Kw - 1) + 1, 300)
        GK.test_depthwise_generic(Kw, stride, pad, dil, Tin)
        return "depthwise Kw={} stride={} pad={} dil={} Tin={}".format(Kw, stride, pad, dil, Tin)
    B, C, T = R.randint(1, 3), R.choice([3, 8, 24, 64]), R.randint(2, 600)
    GK.test_cln_fwd_bwd(B, C, T)
    return "cln fwd+bwd {} {} {}".format(B, C, T)


def case_round4(R):
    """the kernels of round 4: attention core (with and without dropout), gLN over tokens in both forms, the chained cLN with PReLU"""
    pick = R.randint(0, 3)
    if pick == 0:
        N, L, H, D = R.randint(1, 3), R.randint(1, 140), R.randint(1, 3), R.choice([8, 16, 32])
        p = R.choice([0.0, 0.0, 0.3]) if N * L * L * H >= 20000 else 0.0
        GK.test_attention_core_fwd_bwd(N, L, H, D, p)
        return "attention {} {} {} {} {}".format(N, L, H, D, p)
    if pick == 1:
        nseq, L, C = R.randint(1, 6), R.randint(1, 400), R.choice([4, 16, 64, 256, 1024])
        GK.test_gln_tokens_fwd_bwd(nseq, L, C)
        return "gln_tokens {} {} {}".format(nseq, L, C)
    if pick == 2:
        nseq, C = R.randint(1, 3), R.choice([16, 64])
        L = R.randint(16 * 1024 // C * 2, 16 * 1024 // C * 5)           # long enough for the sliced form
        GK.test_gln_tokens_fwd_bwd(nseq, L, C)
        return "gln_tokens sliced {} {} {}".format(nseq, L, C)
    B, C, T, a = R.randint(1, 3), R.choice([5, 24, 64, 100, 200, 300]), R.randint(1, 500), R.choice([0.25, -0.3, 0.0])
    GK.test_prelu_cln_fwd_bwd(B, C, T, a)
    return "prelu_cln {} {} {} {}".format(B, C, T, a)


def case_round5(R):
    """the kernels of round 5: residual sum + LayerNorm over features, ReLU + dropout, the one-pass attention forward at every tile size"""
    pick = R.randint(0, 2)
    if pick == 0:
        C = 4 * R.randint(1, 256)
        rows, res = R.randint(1, 60), R.random() < 0.7
        p = (R.choice([0.0, 0.1, 0.5]) if rows * C >= 4096 else 0.0) if res else 0.0
        GK.test_rownorm_fwd_bwd(rows, C, res, p)
        return "rownorm {} {} {} {}".format(rows, C, res, p)
    if pick == 1:
        n, p = 4 * R.randint(1, 3000), R.choice([0.0, 0.1, 0.3])
        GK.test_relu_drop_fwd_bwd(n, p if n >= 4096 else 0.0)
        return "relu_drop {} {}".format(n, p)
    N, L, H, D = 1, R.choice([1, 31, 32, 33, 64, 65, 100, 128, 129]), R.randint(1, 2), R.choice([8, 16, 32])      # (longer sequences: the fixed cases of the CPU tier)
    GK.test_attention_core_fwd_bwd(N, L, H, D, 0.0)
    return "attention {} {} {} {}".format(N, L, H, D)


CASES = [case_round5, case_round5, case_round4, case_round4, case_gemm, case_gemm, case_gemm_forms, case_gemm_forms, case_wgrad, case_wgrad, case_codec, case_norms, case_chunks, case_lstm, case_dense, case_dense,
         case_test_functions, case_test_functions]


def run_cases(seed, seconds=None, max_cases=None):
    """random cases on whatever backend tests/test_gpu_kernels.py's hooks point at -> (number run, failures)"""
    R = random.Random(seed)
    GK.G.manual_seed(seed)
    failures, n = [], 0
    t_end = time.time() + (seconds or 1e9)
    while time.time() < t_end and (max_cases is None or n + len(failures) < max_cases):
        fn = R.choice(CASES)
        try:
            fn(R)
            n += 1
        except Exception as e:           # noqa: BLE001
            failures.append((fn.__name__, repr(e)[:300], traceback.format_exc(limit=4)))
    return n, failures


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    with tempfile.TemporaryDirectory() as d:
        so = hostsim.build(d)
        with hostsim.HostSimBackend(so) as K:
            GK.HIP, GK.to_device, GK.device_sync, GK.device_name = K, (lambda t: t.clone()), (lambda: None), (lambda: "cpu")
            n, failures = run_cases(seed, seconds=seconds)
    print("{} cases, {} failures (seed {})".format(n, len(failures), seed))
    for f in failures[:10]:
        print(f[0], f[1])
        print(f[2])
    return 1 if failures else 0


if __name__ == "__main__":
    raise SystemExit(main())
