# r05zn: which SDPA backend is fastest at the dual-path separators' attention shapes (fp32, short sequences, narrow heads)?
cd $GRAFT_REPO_ROOT
timeout 600 python /dev/stdin <<'P'
import torch, time
import torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = "cuda"
def run(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, (N, h, L, d) in {"dptnet intra": (257, 4, 250, 16), "dptnet inter": (250, 4, 257, 16), "sepformer": (132, 8, 250, 32), "galr": (128, 8, 81, 8)}.items():
    q, k, v = (torch.randn(N, h, L, d, device=dev, requires_grad=True) for _ in range(3))
    go = torch.randn(N, h, L, d, device=dev)
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
        try:
            with sdpa_kernel(be):
                tf = run(lambda: F.scaled_dot_product_attention(q, k, v))
                def fb():
                    o = F.scaled_dot_product_attention(q, k, v); o.backward(go); q.grad = k.grad = v.grad = None
                tfb = run(fb)
            print("%-14s %-22s fwd %7.1f us   fwd+bwd %7.1f us" % (name, str(be).split('.')[-1], tf, tfb))
        except Exception as e:
            print(name, be, "failed:", str(e)[:80])
P
