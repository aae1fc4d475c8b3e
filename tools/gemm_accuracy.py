"""Development tool: error of sep_pw_gemm against fp64 for operands of very different dynamic range, per arithmetic
(SEPK_GEMM_ARITH = f16x3 | bf16x6 | f32):
    SEPK_GEMM_ARITH=f16x3 python tools/gemm_accuracy.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
import sepkernels  # noqa: E402

K_ = sepkernels.HipBackend()
torch.manual_seed(0)
B, M, T, ldt = 2, 256, 1000, 1024
print("arithmetic:", sepkernels.gemm_arith_name())
for name, Kk, mk in [("unit variance K=128", 128, lambda k: torch.randn(B, k, ldt)),
                     ("unit variance K=512", 512, lambda k: torch.randn(B, k, ldt)),
                     ("channels over e^+-4, K=512", 512, lambda k: torch.randn(B, k, ldt) * torch.exp(4 * torch.randn(1, k, 1))),
                     ("gradient-like 1e-6, K=1024", 1024, lambda k: 1e-6 * torch.randn(B, k, ldt) * torch.exp(2 * torch.randn(1, k, 1))),
                     ("columns over e^+-6, K=512", 512, lambda k: torch.randn(B, k, ldt) * torch.exp(6 * torch.randn(B, 1, ldt))),
                     ("spikes 1e4, K=512", 512, lambda k: torch.randn(B, k, ldt) + 1e4 * (torch.rand(B, k, ldt) < 1e-3)),
                     ("late spike: zeros then 1e3 at k>=496", 512, lambda k: torch.cat([1e-3 * torch.randn(B, k - 16, ldt), 1e3 * torch.randn(B, 16, ldt)], 1))]:
    X = mk(Kk)
    X[..., T:] = 0
    A = torch.randn(M, Kk) * Kk ** -0.5
    ref = torch.einsum("mk,bkt->bmt", A.double(), X.double())[..., :T]
    den = torch.einsum("mk,bkt->bmt", A.double().abs(), X.double().abs())[..., :T]
    Y = torch.full((B, M, ldt), float("nan"), device="cuda")
    K_.pw_gemm(B=B, M=M, K=Kk, T=T, ldt=ldt, A=A.cuda(), X=X.cuda(), Y=Y)
    torch.cuda.synchronize()
    d = (Y.cpu().double()[..., :T] - ref).abs()
    print("  {:40s} max|err|/(|A||X|) {:.2e}   max|err|/max|ref| {:.2e}   finite {}".format(
        name, (d / den.clamp_min(1e-300)).max().item(), d.max().item() / ref.abs().max().item(), bool(torch.isfinite(Y[..., :T]).all())))
