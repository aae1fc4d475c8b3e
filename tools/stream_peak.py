"""Development tool: what plain streaming kernels reach on this part (torch copy / add / sum / fill at 131 MB - 2 GB): the practical
HBM ceiling the depthwise / head / tail kernels are compared with in DESIGN.md (add: 6.0 TB/s, copy 4.8-6.9, fill 6.4-6.9)."""
import torch
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (131, 524, 2096):
    n = mb * 1000 * 1000 // 4
    a, b, c = torch.randn(n, device="cuda"), torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
    ms = t(lambda: c.copy_(a)); print("copy   %4d MB tensors: %.1f us  %.2f TB/s" % (mb, ms * 1e3, 2 * n * 4 / ms / 1e9))
    ms = t(lambda: torch.add(a, b, out=c)); print("add    %4d MB tensors: %.1f us  %.2f TB/s" % (mb, ms * 1e3, 3 * n * 4 / ms / 1e9))
    ms = t(lambda: a.sum()); print("sum    %4d MB tensor : %.1f us  %.2f TB/s" % (mb, ms * 1e3, n * 4 / ms / 1e9))
    ms = t(lambda: c.fill_(1.0)); print("fill   %4d MB tensor : %.1f us  %.2f TB/s" % (mb, ms * 1e3, n * 4 / ms / 1e9))
