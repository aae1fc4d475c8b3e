"""Development tool: per-workgroup timeline and wait statistics of pw_gemm_pcd_kernel (needs a -DPCD_PROF build selected with
SEPKERNELS_LIB).  Wall clock stamps are 100 MHz, cycle counts are s_memtime."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import numpy as np
import torch
import sepkernels
from sepkernels import EPI_RESIDUAL, EPI_ROWSUMS, EPI_ROWSUMS_PRELU, EPI_STATS_PRELU, PRO_GLN, PRO_GLN_PRELU, STATS_SLOTS
K = sepkernels.HipBackend()
lib = ctypes.CDLL(sepkernels.LIB_PATH)
B, T, ldt, H, Bn, Sc, N = 16, 3999, 4096, 512, 128, 128, 512
f = lambda *s: torch.randn(*s, device="cuda")
al = torch.tensor([0.25], device="cuda")
st = lambda: torch.rand(B, STATS_SLOTS, 2, device="cuda", dtype=torch.float64) * 1e3 + torch.tensor([0.0, 1e6], device="cuda", dtype=torch.float64)
cases = {
    "F1": dict(M=Bn, K=N, A=f(Bn, N), X=f(B, N, ldt), Y=f(B, Bn, ldt), bias=f(Bn), pro_mode=PRO_GLN, pro_stats=st(), pro_gamma=f(N), pro_beta=f(N), count=N * T),
    "F3": dict(M=Bn + Sc, K=H, A=f(Bn + Sc, H), X=f(B, H, ldt), Y=f(B, Bn, ldt), Y2=f(B, Sc, ldt), m_split=Bn, bias=f(Bn + Sc), accumulate=1, epi_flags=EPI_RESIDUAL, epi_res=f(B, Bn, ldt),
               pro_mode=PRO_GLN_PRELU, pro_stats=st(), pro_gamma=f(H), pro_beta=f(H), pro_alpha=al, count=H * T),
    "P0": dict(M=H, K=H, A=f(H, H), X=f(B, H, ldt), Y=f(B, H, ldt)),
    "F2": dict(M=H, K=Bn, A=f(H, Bn), X=f(B, Bn, ldt), Y=f(B, H, ldt), bias=f(H), epi_flags=EPI_STATS_PRELU, epi_alpha=al, epi_stats=st()),
    "G3": dict(M=H, K=Bn + Sc, trans_a=1, A=f(Bn, H), A2=f(Sc, H), X=f(B, Bn, ldt), X2=f(B, Sc, ldt), k_split=Bn, Y=f(B, H, ldt),
               epi_flags=EPI_ROWSUMS | EPI_ROWSUMS_PRELU, epi_aux=f(B, H, ldt), epi_alpha=al, epi_rowpart=torch.zeros(B, H, ldt // 64, 2, device="cuda")),
}
for which in sys.argv[1:] or ["F1", "F3", "P0"]:
    kw = dict(cases[which])
    if kw.get("trans_a"):
        W = torch.cat([kw["A"].reshape(-1, kw["M"]), kw["A2"].reshape(-1, kw["M"])], 0).contiguous()
        kw["A_pk"] = K.pack_weights([(W, kw["K"], kw["M"], 1)])[0]
    else:
        kw["A_pk"] = K.pack_weights([(kw["A"], kw["M"], kw["K"], 0)])[0]
    for _ in range(3):
        K.pw_gemm(B=B, T=T, ldt=ldt, eps=1e-12, **kw)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (4096 * 16))()
    assert lib.sep_debug_pcd_prof(buf) == 0
    a = np.array(buf[:]).reshape(4096, 16).astype(np.float64)
    n = int((a[:, 0] > 0).sum())
    a = a[:n]
    nk = kw["K"] // 16
    t0 = a[:, 0].min()
    print("{}: {} workgroups, nk = {}; kernel span {:.1f} us; per workgroup mean us: start->consumer loop end {:.2f} | producer loop end {:.2f} | epilogue {:.2f} | total {:.2f}".format(
        which, n, nk, (a[:, 2].max() - t0) / 100, ((a[:, 1] - a[:, 0]) / 100).mean(), ((a[:, 10] - a[:, 0]) / 100).mean(), ((a[:, 2] - a[:, 1]) / 100).mean(), ((a[:, 2] - a[:, 0]) / 100).mean()))
    print("   consumer wave 0: loop {:.0f} cycles/chunk; sample misses {:.1f} of {} chunks, spin iterations {:.1f}".format(a[:, 5].mean() / nk, a[:, 3].mean(), nk, a[:, 4].mean()))
    print("   producer wave 4: loop {:.0f} cycles/chunk, of which DMA wait {:.0f}; sample misses {:.1f}, spin iterations {:.1f}".format(a[:, 8].mean() / nk, a[:, 9].mean() / nk, a[:, 6].mean(), a[:, 7].mean()))
    if os.environ.get("PCD_PROF2"):
        print("   producer sections, cycles/chunk: raw landed {:.0f} | DMA issue {:.0f} | vmcnt {:.0f} | LDS reads issue {:.0f} (wave 4) | VALU {:.0f} (wave 5) | freed wait {:.0f} | writes+post+sample {:.0f} | loop {:.0f} (wave 6)".format(
            a[:, 11].mean() / nk, a[:, 12].mean() / nk, a[:, 13].mean() / nk, a[:, 14].mean() / nk, a[:, 15].mean() / nk, a[:, 6].mean() / nk, a[:, 7].mean() / nk, a[:, 9].mean() / nk))
    order = np.argsort(a[:, 0])
    print("   start times us (every 64th wg):", np.round((a[order, 0][::64] - t0) / 100, 1))
