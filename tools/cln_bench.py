"""Timing of sep_cln_fwd / sep_cln_bwd (PReLU in front) at the causal paper-best shape (GPU box):  python tools/cln_bench.py"""
import torch

import sepkernels


def run(fn, n=30):
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
    B, C, T, ldt = 16, 512, 3999, 4096
    dev = "cuda"
    x, dy = torch.randn(B, C, ldt, device=dev), torch.randn(B, C, ldt, device=dev)
    gamma, beta, alpha = torch.randn(C, device=dev) + 1, torch.randn(C, device=dev), torch.tensor([0.25], device=dev)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(B, ldt, device=dev), torch.empty(B, ldt, device=dev)
    ws = torch.empty((K.cln_ws_bytes(B, C, T, ldt) + 7) // 8, device=dev, dtype=torch.float64)
    dg, db, da = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
    tf = run(lambda: K.cln_fwd(x, gamma, beta, y, mean, rstd, ws, B, C, T, ldt, 1e-12, alpha=alpha))
    tb = run(lambda: K.cln_bwd(dy, x, gamma, mean, rstd, dx, dg, db, ws, B, C, T, ldt, 1e-12, alpha=alpha, dalpha_part=da))
    mb = B * C * ldt * 4 / 1e6
    print("cLN (PReLU in front) B %d C %d T %d: forward %6.1f us (%4.2f TB/s)   backward %6.1f us (%4.2f TB/s)" % (B, C, T, tf, 2 * mb / tf, tb, 3 * mb / tb))


if __name__ == "__main__":
    main()
