"""Development tool: phase timeline of pw_wgrad_direct_kernel (needs -DSEP_PROF build, see tools/gemm_prof.py).
Stamps of group-0 waves: 0 entry | 1 tables | 2 first chunk landed | 3 loop done | 4 barrier | 5 group-1 handoff | 6 stores acked"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
import sepkernels  # noqa: E402
from sepkernels import PRO_GLN_PRELU, STATS_SLOTS  # noqa: E402
K = sepkernels.HipBackend(); lib = sepkernels.load(); dev = "cuda"
B, T, ldt = 16, 3999, 4096
Bn, H, Sc = 128, 512, 128
f = lambda *s: torch.randn(*s, device=dev)
st = lambda: torch.rand(B, STATS_SLOTS, 2, device=dev, dtype=torch.float64) * 1e3 + torch.tensor([0.0, 1e6], device=dev, dtype=torch.float64)
al = torch.tensor([0.25], device=dev)
xB, xH, xS = f(B, Bn, ldt), f(B, H, ldt), f(B, Sc, ldt)
cases = {"W2 conv1 512x128": dict(M=H, N=Bn, G=xH, X=xB, nsplit=128),
         "W3 heads 256x512 gLNPReLU": dict(M=Bn + Sc, N=H, G=xB, G2=xS, g_split=Bn, X=xH, nsplit=64, x_mode=PRO_GLN_PRELU, x_stats=st(), x_gamma=f(H), x_beta=f(H), x_alpha=al, count=H * T)}
buf = (ctypes.c_longlong * (4 * 4 * 16))()
for name, kw in cases.items():
    part = torch.empty(kw["nsplit"], kw["M"], kw["N"], device=dev); pb = torch.empty(kw["nsplit"], kw["M"], device=dev)
    for _ in range(3):
        K.pw_wgrad(B=B, T=T, ldt=ldt, eps=1e-12, partial=part, partial_bias=pb, **kw)
    torch.cuda.synchronize()
    assert lib.sep_debug_prof(buf) == 0
    print(name)
    for blk in range(4):
        s = [buf[(blk * 4 + 0) * 16 + k] for k in range(7)]
        d = [s[k + 1] - s[k] for k in range(6)]
        print("  blk{} w0: tables {:6d} | 1st chunk {:6d} | loop {:7d} | barrier {:6d} | handoff {:6d} | stores {:6d} cycles (total {:.1f} us)".format(blk, *d, (s[6] - s[0]) / 2400.0))
