"""Timing of sep_attn_fwd / sep_attn_bwd at the dual-path separators' shapes, with and without dropout (GPU box, through gpurun):
    python tools/attn_bench.py [name ...]
fp32 FLOP: forward 4 L^2 D per (sequence, head), backward 10 L^2 D."""
import sys

import torch

import sepkernels

SHAPES = {                                    # N, H, L, D
    "sepformer-intra": (124, 8, 250, 32),
    "sepformer-inter": (1000, 8, 31, 32),
    "dptnet-intra": (255, 4, 250, 16),
    "dptnet-inter": (250, 4, 255, 16),
    "galrnet": (128, 8, 81, 8),
    "dptnet-257": (250, 4, 257, 16),              # one step beyond the 256-step tile (attn_core_ok's limit until round 5)
    "dptnet-320": (250, 4, 320, 16),
}


def run(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    K = sepkernels.HipBackend()
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(SHAPES)
    for name in names:
        N, H, L, D = SHAPES[name]
        qkv = torch.randn(N, L, 3, H, D, device="cuda")
        dout = torch.randn(N, L, H, D, device="cuda")
        o, lse = torch.empty(N, L, H, D, device="cuda"), torch.empty(N, H, L, device="cuda")
        delta, dq = torch.empty_like(lse), torch.empty_like(qkv)
        for pd in (0.0, 0.1):
            tf = run(lambda: K.attn_fwd(qkv, o, lse, N, L, H, D, D ** -0.5, pd, 77))
            tb = run(lambda: K.attn_bwd(qkv, o, dout, lse, delta, dq, N, L, H, D, D ** -0.5, pd, 77))
            gf = N * H * L * L * D * 1e-9
            line = "%-16s N %4d H %d L %3d D %2d p %.1f   fwd %6.1f us (%5.1f TF/s)   bwd %6.1f us (%5.1f TF/s)" % (
                name, N, H, L, D, pd, tf, 4 * gf / tf * 1e3, tb, 10 * gf / tb * 1e3)
            if "--sdpa" in sys.argv:
                import torch.nn.functional as F
                q, k, v = (qkv[:, :, i].transpose(1, 2).contiguous().requires_grad_(True) for i in range(3))
                go = dout.transpose(1, 2).contiguous()
                sf = run(lambda: F.scaled_dot_product_attention(q, k, v, dropout_p=pd))

                def fb():
                    oo = F.scaled_dot_product_attention(q, k, v, dropout_p=pd)
                    oo.backward(go)
                    q.grad = k.grad = v.grad = None
                line += "   | torch SDPA fwd %6.1f fwd+bwd %6.1f us" % (sf, run(fb))
            print(line)


if __name__ == "__main__":
    main()
