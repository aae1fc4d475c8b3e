"""Development tool: times the two depthwise kernels at the paper-best shapes for every dilation of the TCN.
    python tools/stream_bench.py     (SEPKERNELS_LIB / SEPK_DWCONV_LDS=1 select variants)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
import sepkernels  # noqa: E402
from sepkernels import STATS_SLOTS  # noqa: E402

K = sepkernels.HipBackend()
dev = "cuda"
B, C, T, ldt = 16, 512, 3999, int(os.environ.get("LDT", "4096"))
f = lambda *s: torch.randn(*s, device=dev)
st = lambda: torch.rand(B, STATS_SLOTS, 2, device=dev, dtype=torch.float64) * 1e3 + torch.tensor([0.0, 1e6], device=dev, dtype=torch.float64)
a, z, dv2, dv1 = f(B, C, ldt), f(B, C, ldt), f(B, C, ldt), f(B, C, ldt)
g1, b1, g2 = f(C), f(C), f(C)
a1, a2 = torch.tensor([0.25], device=dev), torch.tensor([0.1], device=dev)
wd, bd = f(C, 1, 3), f(C)
st1, st2 = st(), st()
bsum2, bacc1, arrive1, bsum1 = f(B, 2) * 0.01, st(), torch.zeros(B, 17, device=dev, dtype=torch.int32), f(B, 2)
rp = torch.empty(B, C, (ldt + 1023) // 1024, 8, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


H = B * C * ldt * 4 / 1e6   # MB per H-tensor
tf = tb = 0.0
for d in (1, 2, 4, 8, 16, 32, 64, 128):
    ms_f = timeit(lambda: K.dwconv_fwd(a, st1, g1, b1, a1, wd, bd, a2, z, st2, B, C, T, ldt, d, 1e-12))
    mode = os.environ.get("DWB_MODE", "sums")      # none | sums (what the step uses) | publish
    extra = {"none": (None, None, None), "sums": (bacc1, None, None), "publish": (bacc1, arrive1, bsum1)}[mode]
    ms_b = timeit(lambda: K.dwconv_bwd(dv2, z, a, st1, g1, b1, a1, st2, g2, a2, bsum2, wd, (None if os.environ.get("DWB_READ_Z") else bd), dv1, rp, *extra, B, C, T, ldt, d, 1e-12))
    tf += ms_f
    tb += ms_b
    print("d={:4d}  fwd {:7.1f} us ({:5.2f} TB/s)   bwd {:7.1f} us ({:5.2f} TB/s)".format(d, 1e3 * ms_f, 2 * H / ms_f / 1e3, 1e3 * ms_b, 4 * H / ms_b / 1e3))
print("per 8 layers: fwd {:.3f} ms, bwd {:.3f} ms".format(tf, tb))
