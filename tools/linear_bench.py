"""Timing of sep_linear_fwd / bwd_input / bwd_weight at the dual-path separators' shapes (GPU box): us per call and fp32 TFLOP/s.
    python tools/linear_bench.py"""
import torch

import sepkernels

SHAPES = {                                   # tokens, in, out
    "sepformer qkv 33k x 256 -> 768": (33000, 256, 768),
    "sepformer out 33k x 256 -> 256": (33000, 256, 256),
    "dprnn gates 128k x 64 -> 1024": (128500, 64, 1024),
    "dprnn fc 128k x 256 -> 64": (128500, 256, 64),
    "dptnet qkv 64k x 64 -> 192": (64250, 64, 192),
    "dptnet gates 64k x 64 -> 1024": (64250, 64, 1024),
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
    dev = "cuda"
    for name, (ntok, Kin, N) in SHAPES.items():
        x = torch.randn(ntok, Kin, device=dev)
        w = torch.randn(N, Kin, device=dev) * Kin ** -0.5
        b = torch.randn(N, device=dev)
        y, dy, dx = torch.empty(ntok, N, device=dev), torch.randn(ntok, N, device=dev), torch.empty(ntok, Kin, device=dev)
        ns = max(1, min(512 // max(1, (N // (128 if N % 128 == 0 else 64)) * (Kin // (128 if Kin % 128 == 0 else 64))), (ntok + 255) // 256))
        part, pb = torch.empty(ns, N, Kin, device=dev), torch.empty(ns, N, device=dev)
        tf = run(lambda: K.linear_fwd(x, w, b, None, y, ntok, Kin, N))
        ti = run(lambda: K.linear_bwd_input(dy, w, dx, ntok, Kin, N, 0))
        tw = run(lambda: K.linear_bwd_weight(dy, x, Kin, part, pb, ntok, Kin, N, 1, 0, ns))
        gf = 2e-9 * ntok * Kin * N
        print("%-34s fwd %6.1f us (%5.1f TF/s)  d input %6.1f (%5.1f)  d weight %6.1f (%5.1f)" % (name, tf, gf / tf * 1e3, ti, gf / ti * 1e3, tw, gf / tw * 1e3))


if __name__ == "__main__":
    main()
