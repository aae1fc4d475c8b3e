"""diagnostic: sep_dwconv_bwd direct on many rows vs the emulator; where do the outputs differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "dnn-based_source_separation_amd", "src")):
    sys.path.insert(0, p)
import torch
import test_gpu_kernels as GK
from test_gpu_kernels import *  # noqa
B, C, T, d = [int(v) for v in sys.argv[1:5]]
ldt = (T + 127) // 128 * 128
a = padded(B, C, T, ldt)
a1, a2 = torch.tensor([0.25]), torch.tensor([0.1])
u1 = torch.where(a > 0, a, a1 * a)
st1 = stats_of(u1, T)
g1, b1, wd, bd = rnd(C) + 1, rnd(C), rnd(C, 1, 3), rnd(C)
z, st2 = nan(B, C, ldt), zstats(B)
EMU.dwconv_fwd(a, st1, g1, b1, a1, wd, bd, a2, z, st2, B, C, T, ldt, d, 1e-12)
ntile = (ldt + 1023) // 1024
args = [padded(B, C, T, ldt), z, a, st1, g1, b1, a1, st2, rnd(C) + 1, a2, rnd(B, 2, scale=0.01), wd, bd, nan(B, C, ldt), nan(B, C, ntile, 8), zstats(B),
        torch.zeros(B, 17, dtype=torch.int32), None, B, C, T, ldt, d, 1e-12]
gargs = [to_device(v) if torch.is_tensor(v) else v for v in args]
xpub = to_device(torch.full((B, 8, 2), float("nan"), dtype=torch.float64))
EMU.dwconv_bwd(*args, direct=1)
HIP.dwconv_bwd(*gargs, direct=1, xpub=xpub)
device_sync()
print("timeouts", HIP.sync_timeouts(), "max rows", HIP.dwconv_bwd_direct_max_rows(ldt, True))
ref, got = args[13], gargs[13].cpu()
err = (ref - got).abs().amax(2)            # (B, C)
scale = ref.abs().max()
print("max rel err", (err.max() / scale).item())
bad = (err > 2e-4 * scale).nonzero()
print("bad rows", bad.shape[0], "of", B * C)
if bad.shape[0]:
    rows = bad[:, 0] * C + bad[:, 1]
    print("first bad rows", rows[:20].tolist())
    print("bad rows per sample", [(bad[:, 0] == b).sum().item() for b in range(B)])
    print("bad row % 8 histogram", torch.bincount(rows % 8, minlength=8).tolist())
    print("bad row // 1024 histogram", torch.bincount(rows // 1024).tolist())
xs = xpub.cpu()
bc = args[15].sum(1)
print("published totals vs emulator (per sample):")
for b in range(B):
    print(b, xs[b].nan_to_num(0).sum(0).tolist(), bc[b].tolist(), "nan entries", int(xs[b].isnan().sum()))
print("arrive", gargs[16].cpu()[:, :8].tolist())
rc, rg = args[14].double().sum(2), gargs[14].cpu().double().sum(2)
print("rowpart rel err per slot", [((rc[..., i] - rg[..., i]).abs().max() / (rc[..., i].abs().max() + 1e-30)).item() for i in range(8)])
