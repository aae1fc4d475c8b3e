"""Development tool: per-step phase timeline of pw_gemm_coop_kernel from s_memtime stamps (needs a -DCOOP_PROF build of
gemm_coop.hip, selected with SEPKERNELS_LIB).  Stamps per chunk step of wave 0 of four sampled workgroups:
0 step start | 1 operands + raw X arrived (forced lgkmcnt(0)) | 2 MFMAs + split issued | 3 own DMA + ds_write done | 4 barrier passed | 5 DMA group issued."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch
import sepkernels
from sepkernels import EPI_RESIDUAL, PRO_GLN_PRELU, STATS_SLOTS
K = sepkernels.HipBackend()
lib = ctypes.CDLL(sepkernels.LIB_PATH)
B, T, ldt, H, Bn, Sc = 16, 3999, 4096, 512, 128, 128
f = lambda *s: torch.randn(*s, device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "F3"
if which == "F3":
    A = f(Bn + Sc, H)
    kw = dict(M=Bn + Sc, K=H, A=A, X=f(B, H, ldt), Y=f(B, Bn, ldt), Y2=f(B, Sc, ldt), m_split=Bn, bias=f(Bn + Sc), accumulate=1, epi_flags=EPI_RESIDUAL, epi_res=f(B, Bn, ldt),
              pro_mode=PRO_GLN_PRELU, pro_stats=torch.rand(B, STATS_SLOTS, 2, device="cuda", dtype=torch.float64) * 1e3 + torch.tensor([0.0, 1e6], device="cuda", dtype=torch.float64),
              pro_gamma=f(H), pro_beta=f(H), pro_alpha=torch.tensor([0.25], device="cuda"), count=H * T)
else:
    A = f(H, H)
    kw = dict(M=H, K=H, A=A, X=f(B, H, ldt), Y=f(B, H, ldt))
kw["A_pk"] = K.pack_weights([(A, kw["M"], kw["K"], 0)])[0]
for _ in range(3):
    K.pw_gemm(B=B, T=T, ldt=ldt, eps=1e-12, **kw)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (4 * 64 * 8))()
assert lib.sep_debug_coop_prof(buf) == 0
import numpy as np
a = np.array(buf[:]).reshape(4, 64, 8)
nk = kw["K"] // 16
for s in range(4):
    st = a[s, :nk - 1, :6]
    d = np.diff(st, axis=1)
    step = np.diff(a[s, :nk - 1, 0])
    print("block sample", s, "mean cycles per phase [reads, mfma+split issue, own waits, barrier, dma issue]:", d[2:-2].mean(0).round(0), " step:", step[2:-2].mean().round(0))
