"""Development tool: per-chunk role timeline of pw_wgrad_pc16_kernel (needs a -DWPC16_PROF build selected with SEPKERNELS_LIB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import numpy as np
import torch
import sepkernels
from sepkernels import PRO_PRELU
K = sepkernels.HipBackend()
lib = ctypes.CDLL(sepkernels.LIB_PATH)
B, T, ldt, H, Bn, Sc = 16, 3999, 4096, 512, 128, 128
f = lambda *s: torch.randn(*s, device="cuda")
for name, kw, ns, M, N in (("heads 256x512 PReLU", dict(M=Bn + Sc, N=H, G=f(B, Bn, ldt), G2=f(B, Sc, ldt), g_split=Bn, X=f(B, H, ldt), x_mode=PRO_PRELU, x_alpha=torch.tensor([0.25], device="cuda")), 64, 256, 512),
                           ("conv1 512x128", dict(M=H, N=Bn, G=f(B, H, ldt), X=f(B, Bn, ldt)), 128, 512, 128)):
    part, pb = torch.empty(ns, M, N, device="cuda"), torch.empty(ns, M, device="cuda")
    for _ in range(3):
        K.pw_wgrad(B=B, T=T, ldt=ldt, eps=1e-12, partial=part, partial_bias=pb, nsplit=ns, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.pw_wgrad(B=B, T=T, ldt=ldt, eps=1e-12, partial=part, partial_bias=pb, nsplit=ns, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    sb = (ctypes.c_longlong * (2 * 64 * 8))()
    assert lib.sep_debug_wpc_step(sb) == 0
    s = np.array(sb[:]).reshape(2, 64, 8).astype(np.float64)
    nkk = 4096 * 16 // 16 // ns
    pr = s[1, 4:min(60, nkk - 2), :5]
    nst = min(64, nkk)
    span = s[0, nst - 2 if nst % 2 == 0 else nst - 1, 0] - s[0, 0, 0]
    print(name, "chunks per slab", nkk, " kernel {:.1f} us;  consumer stamps chunk 0 -> {}: {:.0f} ticks ({:.0f} per chunk)".format(us, nst - 2, span, span / max(1, nst - 2)))
    print("  producer cycles: read+split+write {:.0f} | lgkm drain {:.0f} | vm wait + barrier {:.0f} | DMA issue {:.0f} (every other chunk) | step {:.0f}".format(
        (pr[:, 1] - pr[:, 0]).mean(), (pr[:, 2] - pr[:, 1]).mean(), (pr[:, 3] - pr[:, 2]).mean(), (pr[:, 4] - pr[:, 3]).mean(), np.diff(pr[:, 0]).mean()))
    co = s[0, 4:min(60, nkk - 2):2, :5]
    print("  consumer cycles (two chunks): barrier {:.0f} | step {:.0f} | barrier {:.0f} | step {:.0f} | pair {:.0f}".format(
        (co[:, 1] - co[:, 0]).mean(), (co[:, 2] - co[:, 1]).mean(), (co[:, 3] - co[:, 2]).mean(), (co[:, 4] - co[:, 3]).mean(), np.diff(co[:, 0]).mean()))
