"""Development tool: per-workgroup timeline of pw_gemm_pc_kernel (needs a -DPC_PROF build selected with SEPKERNELS_LIB).
Stamps (wall clock, 100 MHz): 0 start | 1 rings filled (producer, after B_-1) | 2 main loop done (consumer 0) | 4 (producer) | 3 epilogue done."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import numpy as np
import torch
import sepkernels
from sepkernels import EPI_RESIDUAL, EPI_ROWSUMS, EPI_ROWSUMS_PRELU, EPI_STATS_PRELU, PRO_GLN_PRELU, STATS_SLOTS
K = sepkernels.HipBackend()
lib = ctypes.CDLL(sepkernels.LIB_PATH)
B, T, ldt, H, Bn, Sc = 16, 3999, 4096, 512, 128, 128
f = lambda *s: torch.randn(*s, device="cuda")
al = torch.tensor([0.25], device="cuda")
st = lambda: torch.rand(B, STATS_SLOTS, 2, device="cuda", dtype=torch.float64) * 1e3 + torch.tensor([0.0, 1e6], device="cuda", dtype=torch.float64)
cases = {
    "F2": dict(M=H, K=Bn, A=f(H, Bn), X=f(B, Bn, ldt), Y=f(B, H, ldt), bias=f(H), epi_flags=EPI_STATS_PRELU, epi_alpha=al, epi_stats=st()),
    "F3": dict(M=Bn + Sc, K=H, A=f(Bn + Sc, H), X=f(B, H, ldt), Y=f(B, Bn, ldt), Y2=f(B, Sc, ldt), m_split=Bn, bias=f(Bn + Sc), accumulate=1, epi_flags=EPI_RESIDUAL, epi_res=f(B, Bn, ldt),
               pro_mode=PRO_GLN_PRELU, pro_stats=st(), pro_gamma=f(H), pro_beta=f(H), pro_alpha=al, count=H * T),
    "P0": dict(M=H, K=H, A=f(H, H), X=f(B, H, ldt), Y=f(B, H, ldt)),
}
for which in sys.argv[1:] or ["F2", "F3", "P0"]:
    kw = dict(cases[which])
    kw["A_pk"] = K.pack_weights([(kw["A"], kw["M"], kw["K"], 0)])[0]
    for _ in range(3):
        K.pw_gemm(B=B, T=T, ldt=ldt, eps=1e-12, **kw)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (4096 * 8))()
    assert lib.sep_debug_pc_prof(buf) == 0
    a = np.array(buf[:]).reshape(4096, 8).astype(np.float64)
    n = int((a[:, 0] > 0).sum())
    a = a[:n]
    t0 = a[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    fill, loop, epi = (a[:, 1] - a[:, 0]) / 100, (a[:, 2] - a[:, 1]) / 100, (a[:, 3] - a[:, 2]) / 100
    print("{}: {} workgroups; kernel span {:.1f} us; per workgroup mean us: fill {:.2f}  main loop {:.2f}  epilogue {:.2f}  total {:.2f}   (nk = {})".format(
        which, n, us(a[:, 3].max()), fill.mean(), loop.mean(), epi.mean(), ((a[:, 3] - a[:, 0]) / 100).mean(), kw["K"] // 16))
    order = np.argsort(a[:, 0])
    starts = us(a[order, 0])
    print("   start times us (every 128th wg):", np.round(starts[::128], 1))
    sb = (ctypes.c_longlong * (2 * 64 * 8))()
    if hasattr(lib, "sep_debug_pc_step") and lib.sep_debug_pc_step(sb) == 0:
        s = np.array(sb[:]).reshape(2, 64, 8).astype(np.float64)
        nk = kw["K"] // 16
        pr = s[1, 2:nk - 2, :5]
        print("   producer (wave 5) cycles: [B exit->next loop top: n/a] split+write {:.0f} | own DMA wait {:.0f} | barrier wait {:.0f} | DMA issue {:.0f} | step {:.0f}".format(
            (pr[:, 1] - pr[:, 0]).mean(), (pr[:, 2] - pr[:, 1]).mean(), (pr[:, 3] - pr[:, 2]).mean(), (pr[:, 4] - pr[:, 3]).mean(), np.diff(pr[:, 0]).mean()))
        co = s[0, 2:nk - 2:2, :5]
        print("   consumer (wave 1) cycles per PAIR of chunks: barrier wait {:.0f} | reads+24 MFMA issue {:.0f} | barrier wait {:.0f} | reads+24 MFMA issue {:.0f} | pair {:.0f}".format(
            (co[:, 1] - co[:, 0]).mean(), (co[:, 2] - co[:, 1]).mean(), (co[:, 3] - co[:, 2]).mean(), (co[:, 4] - co[:, 3]).mean(), np.diff(co[:, 0]).mean()))
