"""Development tool: error of the stand-alone f16x3 / bf16x6 GEMM paths relative to the fp32-MFMA path over 12 operand draws per
shape (the distribution behind the gate of tests/test_gpu_kernels.py::test_gemm_split_arithmetic_is_as_accurate_as_fp32_mfma)."""
import sys
sys.path.insert(0, "dnn-based_source_separation_amd/src")
import torch, sepkernels
HIP = sepkernels.HipBackend()
for K, scale in [(128, 1.0), (512, 1.0), (1024, 1e-3), (512, 1e4)]:
    rat = []
    for seed in range(12):
        g = torch.Generator().manual_seed(seed)
        B, M, T, ldt = 2, 256, 1000, 1024
        X = torch.zeros(B, K, ldt); X[..., :T] = torch.randn(B, K, T, generator=g)
        X = X * torch.exp(3 * torch.randn(B, K, 1, generator=g))
        A = torch.randn(M, K, generator=g) * scale
        ref = torch.einsum("mk,bkt->bmt", A.double(), X.double())
        err = {}
        for name in ("f32", "bf16x6", "f16x3"):
            Y = torch.full((B, M, ldt), float("nan"), device="cuda")
            HIP.pw_gemm(B=B, M=M, K=K, T=T, ldt=ldt, A=A.cuda(), X=X.cuda(), Y=Y, arith=sepkernels.arith_code(name))
            torch.cuda.synchronize()
            d = (Y.cpu().double() - ref)[..., :T]
            err[name] = (d.abs().max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
        rat.append((err["f16x3"][0] / err["f32"][0], err["f16x3"][1] / err["f32"][1], err["bf16x6"][1] / err["f32"][1]))
    print(K, scale, "max-ratio f16x3 (inf, rms):", round(max(r[0] for r in rat), 2), round(max(r[1] for r in rat), 2), " median rms:", round(sorted(r[1] for r in rat)[6], 2), " bf16x6 rms max:", round(max(r[2] for r in rat), 2))
