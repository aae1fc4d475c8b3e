"""Development tool (not part of the product or the tests): times the MFMA kernels on the exact shapes of the
paper-best training step, one shape at a time, so that kernel variants can be A/B-ed in seconds on the GPU box.
    python tools/gemm_bench.py [--reps 20]          (SEPKERNELS_LIB=/path/to/variant.so selects a build)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
import sepkernels  # noqa: E402
from sepkernels import (EPI_PRELU_BWD, EPI_RESIDUAL, EPI_ROWSUMS, EPI_ROWSUMS_PRELU, EPI_SIGMOID, EPI_STATS_PRELU,  # noqa: E402
                        PRO_GLN, PRO_GLN_BWD, PRO_GLN_PRELU, PRO_PRELU, STATS_SLOTS)

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--ldt", type=int, default=4096)
ap.add_argument("--nsplit-scale", type=float, default=1.0)
ap.add_argument("--packed", action="store_true", help="hand sep_pw_gemm the weights pre-split by sep_pack_weights (A_pk)")
args = ap.parse_args()
K = sepkernels.HipBackend()
dev = "cuda"
B, T, ldt = 16, 3999, args.ldt
N, Bn, H, Sc, ns = 512, 128, 512, 128, 2
f = lambda *s: torch.randn(*s, device=dev)
z = lambda *s: torch.zeros(*s, device=dev)
st = lambda: torch.rand(B, STATS_SLOTS, 2, device=dev, dtype=torch.float64) * 1e3 + torch.tensor([0.0, 1e6], device=dev, dtype=torch.float64)
al = torch.tensor([0.25], device=dev)
d64 = lambda: torch.zeros(1, device=dev, dtype=torch.float64)
xN, xB, xH, xS, xM = f(B, N, ldt), f(B, Bn, ldt), f(B, H, ldt), f(B, Sc, ldt), f(B, ns * N, ldt)

cases = {
    "F1 bottleneck  M128 K512 gLN-pro": dict(M=Bn, K=N, A=f(Bn, N), X=xN, Y=z(B, Bn, ldt), bias=f(Bn), pro_mode=PRO_GLN, pro_stats=st(), pro_gamma=f(N), pro_beta=f(N), count=N * T),
    "F2 conv1       M512 K128 stats-epi": dict(M=H, K=Bn, A=f(H, Bn), X=xB, Y=z(B, H, ldt), bias=f(H), epi_flags=EPI_STATS_PRELU, epi_alpha=al, epi_stats=st()),
    "F3 heads       M256 K512 gLNPReLU-pro res+acc": dict(M=Bn + Sc, K=H, A=f(Bn + Sc, H), X=xH, Y=z(B, Bn, ldt), Y2=z(B, Sc, ldt), m_split=Bn, bias=f(Bn + Sc), accumulate=1, epi_flags=EPI_RESIDUAL, epi_res=xB, pro_mode=PRO_GLN_PRELU, pro_stats=st(), pro_gamma=f(H), pro_beta=f(H), pro_alpha=al, count=H * T),
    "F4 mask        M1024 K128 PReLU-pro sigmoid": dict(M=ns * N, K=Sc, A=f(ns * N, Sc), X=xS, Y=z(B, ns * N, ldt), bias=f(ns * N), pro_mode=PRO_PRELU, pro_alpha=al, epi_flags=EPI_SIGMOID),
    "G4 mask dgrad  M128 K1024 T PReLU-bwd": dict(M=Sc, K=ns * N, trans_a=1, A=f(ns * N, Sc), X=xM, Y=z(B, Sc, ldt), epi_flags=EPI_PRELU_BWD, epi_aux=xS, epi_alpha=al, epi_dalpha=d64()),
    "G3 heads dgrad M512 K256 T rowsums": dict(M=H, K=Bn + Sc, trans_a=1, A=f(Bn, H), A2=f(Sc, H), X=xB, X2=xS, k_split=Bn, Y=z(B, H, ldt), epi_flags=EPI_ROWSUMS | EPI_ROWSUMS_PRELU, epi_aux=xH, epi_alpha=al, epi_rowpart=z(B, H, ldt // 64, 2)),
    "G3p heads dgrad M512 K256 T plain": dict(M=H, K=Bn + Sc, trans_a=1, A=f(Bn, H), A2=f(Sc, H), X=xB, X2=xS, k_split=Bn, Y=z(B, H, ldt)),
    "G2 conv1 dgrad M128 K512 T gLN-bwd-pro res": dict(M=Bn, K=H, trans_a=1, A=f(H, Bn), X=f(B, H, ldt), Y=z(B, Bn, ldt), pro_mode=PRO_GLN_BWD, pro_stats=st(), pro_gamma=f(H), pro_alpha=al, pro_aux=xH, pro_bsum=f(B, 2) * 0.01, pro_store=z(B, H, ldt), pro_dalpha=d64(), count=H * T, epi_flags=EPI_RESIDUAL, epi_res=xB),
    "G1 bneck dgrad M512 K128 T rowsums": dict(M=N, K=Bn, trans_a=1, A=f(Bn, N), X=xB, Y=z(B, N, ldt), epi_flags=EPI_ROWSUMS, epi_aux=xN, epi_rowpart=z(B, N, ldt // 64, 2)),
    "P0 plain       M512 K512": dict(M=H, K=H, A=f(H, H), X=xH, Y=z(B, H, ldt)),
    "P1 plain       M128 K128": dict(M=Bn, K=Bn, A=f(Bn, Bn), X=xB, Y=z(B, Bn, ldt)),
}
wcases = {
    "W2 conv1 wgrad   512x128": dict(M=H, N=Bn, G=xH, X=xB, nsplit=128),
    "W3 heads wgrad   256x512 gLNPReLU": dict(M=Bn + Sc, N=H, G=xB, G2=xS, g_split=Bn, X=xH, nsplit=64, x_mode=PRO_GLN_PRELU, x_stats=st(), x_gamma=f(H), x_beta=f(H), x_alpha=al, count=H * T),
    "W3u heads wgrad  256x512 PReLU aligned": dict(M=Bn + Sc, N=H, G=xB, G2=xS, g_split=Bn, X=xH, nsplit=64, x_mode=PRO_PRELU, x_alpha=al),
    "W3p heads wgrad  256x512 PReLU aligned, dS pre-split": dict(M=Bn + Sc, N=H, G=xB, G2=xS, g_split=Bn, X=xH, nsplit=64, x_mode=PRO_PRELU, x_alpha=al, G2_pre="split"),
    "W4 mask wgrad   1024x128 PReLU": dict(M=ns * N, N=Sc, G=xM, X=xS, nsplit=64, x_mode=PRO_PRELU, x_alpha=al),
    "W1 bneck wgrad   128x512 gLN": dict(M=Bn, N=N, G=xB, X=xN, nsplit=128, x_mode=PRO_GLN, x_stats=st(), x_gamma=f(N), x_beta=f(N), count=N * T),
    "WE enc wgrad     512x16": dict(M=N, N=16, G=xN, X=f(B, 16, ldt), nsplit=128),
}


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


tot = 0.0
for name, kw in cases.items():
    if args.only and not any(name.startswith(o) for o in args.only.split(",")):
        continue
    if args.packed:
        M_, K_ = kw["M"], kw["K"]
        if kw.get("trans_a"):
            W = kw["A"].reshape(-1, M_) if kw.get("A2") is None else torch.cat([kw["A"].reshape(-1, M_), kw["A2"].reshape(-1, M_)], 0).contiguous()
            kw = dict(kw, A_pk=K.pack_weights([(W, K_, M_, 1)])[0])
        else:
            kw = dict(kw, A_pk=K.pack_weights([(kw["A"], M_, K_, 0)])[0])
    ms = timeit(lambda: K.pw_gemm(B=B, T=T, ldt=ldt, eps=1e-12, **kw))
    fl = 2.0 * kw["M"] * kw["K"] * B * T
    rows = kw["K"] + kw["M"] + (kw["M"] - kw.get("m_split", 0) if kw.get("accumulate") else 0) + ((kw.get("m_split", 0) or kw["M"]) if kw.get("epi_res") is not None else 0) \
        + (kw["M"] if kw.get("epi_aux") is not None else 0) + (2 * kw["K"] if kw.get("pro_store") is not None else 0)
    by = 4.0 * rows * B * ldt
    print("{:50s} {:8.1f} us  {:6.1f} TF/s-eq  {:5.2f} TB/s algorithmic ({:4.1f}% of 6.3)".format(name, 1e3 * ms, fl / ms / 1e9, by / ms / 1e9, by / ms / 1e9 / 0.063))
for name, kw in wcases.items():
    if args.only and not any(name.startswith(o) for o in args.only.split(",")):
        continue
    kw = dict(kw, nsplit=int(round(kw["nsplit"] * args.nsplit_scale)))
    ns_ = kw["nsplit"]
    if kw.get("G2_pre") == "split":
        kw["G2_pre"] = K.split_rows(kw["G2"], T, ns_ // B)
    part = torch.empty(ns_, kw["M"], kw["N"], device=dev)
    pb = torch.empty(ns_, kw["M"], device=dev)
    ms = timeit(lambda: K.pw_wgrad(B=B, T=T, ldt=ldt, eps=1e-12, partial=part, partial_bias=pb, **kw))
    fl = 2.0 * kw["M"] * kw["N"] * B * T
    print("{:50s} {:8.1f} us  {:6.1f} TF/s  ({:4.1f}% of 157.3)".format(name, 1e3 * ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573))
