"""Development tool: phase timeline of pw_gemm_direct_kernel from in-kernel s_memtime stamps (needs a -DSEP_PROF build:
SEPKERNELS_LIB=variants/lib_PROF.so python tools/gemm_prof.py).  Stamps per wave of 4 sampled workgroups:
0 entry | 1 prologue tables done | 2 first chunk landed | 3 main loop done | 4 post-loop barrier | 5 epilogue done | 6 stores acked"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
import sepkernels  # noqa: E402
from sepkernels import EPI_ROWSUMS, EPI_ROWSUMS_PRELU, EPI_SIGMOID, EPI_STATS_PRELU, PRO_PRELU, STATS_SLOTS  # noqa: E402

K = sepkernels.HipBackend()
lib = sepkernels.load()
dev = "cuda"
B, T, ldt = 16, 3999, 4096
N, Bn, H, Sc, ns = 512, 128, 512, 128, 2
f = lambda *s: torch.randn(*s, device=dev)
z = lambda *s: torch.zeros(*s, device=dev)
st = lambda: torch.rand(B, STATS_SLOTS, 2, device=dev, dtype=torch.float64) * 1e3
al = torch.tensor([0.25], device=dev)
xB, xH, xS = f(B, Bn, ldt), f(B, H, ldt), f(B, Sc, ldt)
cases = {
    "F2 conv1 M512 K128 stats": dict(M=H, K=Bn, A=f(H, Bn), X=xB, Y=z(B, H, ldt), bias=f(H), epi_flags=EPI_STATS_PRELU, epi_alpha=al, epi_stats=st()),
    "F4 mask M1024 K128 sigmoid": dict(M=ns * N, K=Sc, A=f(ns * N, Sc), X=xS, Y=z(B, ns * N, ldt), bias=f(ns * N), pro_mode=PRO_PRELU, pro_alpha=al, epi_flags=EPI_SIGMOID),
    "G3 heads dgrad M512 K256 rowsums": dict(M=H, K=Bn + Sc, trans_a=1, A=f(Bn, H), A2=f(Sc, H), X=xB, X2=xS, k_split=Bn, Y=z(B, H, ldt), epi_flags=EPI_ROWSUMS | EPI_ROWSUMS_PRELU, epi_aux=xH, epi_alpha=al, epi_rowpart=z(B, H, ldt // 64, 2)),
    "P0 plain M512 K512": dict(M=H, K=H, A=f(H, H), X=xH, Y=z(B, H, ldt)),
}
buf = (ctypes.c_longlong * (4 * 4 * 16))()
for name, kw in cases.items():
    for _ in range(3):
        K.pw_gemm(B=B, T=T, ldt=ldt, eps=1e-12, **kw)
    torch.cuda.synchronize()
    assert lib.sep_debug_prof(buf) == 0
    print(name)
    for blk in range(4):
        for w in range(4):
            s = [buf[(blk * 4 + w) * 16 + k] for k in range(16)]
            d = [s[k + 1] - s[k] for k in range(6)]
            print("  blk{} w{}: tables {:6d} | 1st chunk {:6d} | loop {:7d} | barrier {:5d} | epilogue {:6d} | store ack {:6d}  cycles (total {:.1f} us @2.4GHz)".format(
                blk, w, *d, (s[6] - s[0]) / 2400.0))
            print("           epilogue detail: half0 transpose {:6d} | loads {:6d} | compute+stores {:6d} || half1 transpose {:6d} | loads {:6d} | compute+stores {:6d}".format(
                s[8] - s[4], s[9] - s[8], s[10] - s[9], s[12] - s[10], s[13] - s[12], s[14] - s[13]))

    # workgroup residency: how many workgroups are alive at once (start/end stamps of every workgroup, 100 MHz wall clock)
    n = 8 * ((kw["M"] + 127) // 128) * ((B * (ldt // 128) + 7) // 8)
    st_ = (ctypes.c_longlong * 8192)()
    en_ = (ctypes.c_longlong * 8192)()
    assert lib.sep_debug_blocks(st_, en_) == 0
    n = min(n, 8192)
    ev = sorted([(st_[i], 1) for i in range(n)] + [(en_[i], -1) for i in range(n)])
    t0_, live, peak, area, last = ev[0][0], 0, 0, 0, ev[0][0]
    for tt, dlt in ev:
        area += live * (tt - last)
        last = tt
        live += dlt
        peak = max(peak, live)
    dur = [(en_[i] - st_[i]) / 100.0 for i in range(n)]
    print("  workgroups {}: peak alive {} ({:.2f} per CU), mean alive {:.0f}, span {:.1f} us, mean workgroup life {:.1f} us".format(
        n, peak, peak / 256.0, area / max(last - t0_, 1), (last - t0_) / 100.0, sum(dur) / n))
