"""Development tool: per-chunk role timeline of pw_wgrad_pc_kernel (needs a -DWPC_PROF build selected with SEPKERNELS_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import numpy as np
import torch
import sepkernels
from sepkernels import PRO_GLN_PRELU, STATS_SLOTS
K = sepkernels.HipBackend()
lib = ctypes.CDLL(sepkernels.LIB_PATH)
B, T, ldt, H, Bn, Sc = 16, 3999, 4096, 512, 128, 128
f = lambda *s: torch.randn(*s, device="cuda")
st = torch.rand(B, STATS_SLOTS, 2, device="cuda", dtype=torch.float64) * 1e3 + torch.tensor([0.0, 1e6], device="cuda", dtype=torch.float64)
kw = dict(M=Bn + Sc, N=H, G=f(B, Bn, ldt), G2=f(B, Sc, ldt), g_split=Bn, X=f(B, H, ldt), nsplit=64, x_mode=PRO_GLN_PRELU, x_stats=st, x_gamma=f(H), x_beta=f(H),
          x_alpha=torch.tensor([0.25], device="cuda"), count=H * T)
part, pb = torch.empty(64, 256, 512, device="cuda"), torch.empty(64, 256, device="cuda")
for _ in range(3):
    K.pw_wgrad(B=B, T=T, ldt=ldt, eps=1e-12, partial=part, partial_bias=pb, **kw)
torch.cuda.synchronize()
sb = (ctypes.c_longlong * (2 * 64 * 8))()
assert lib.sep_debug_wpc_step(sb) == 0
s = np.array(sb[:]).reshape(2, 64, 8).astype(np.float64)
pr = s[1, 4:60, :5]
print("producer cycles: read+split+write {:.0f} | own DMA wait {:.0f} | barrier wait {:.0f} | DMA issue {:.0f} | step {:.0f}".format(
    (pr[:, 1] - pr[:, 0]).mean(), (pr[:, 2] - pr[:, 1]).mean(), (pr[:, 3] - pr[:, 2]).mean(), (pr[:, 4] - pr[:, 3]).mean(), np.diff(pr[:, 0]).mean()))
co = s[0, 4:60, :3]
print("consumer cycles: barrier wait {:.0f} | reads + 48 MFMA issue {:.0f} | step {:.0f}".format((co[:, 1] - co[:, 0]).mean(), (co[:, 2] - co[:, 1]).mean(), np.diff(co[:, 0]).mean()))
