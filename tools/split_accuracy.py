"""Development tool (CPU, no GPU needed): error models of fp32-GEMM emulations on narrow matrix-core formats, against fp64.
    python tools/split_accuracy.py
  bf16x6   what SEP_ARITH_BF16X6 does: exact three-way truncated-bf16 split of both operands, 6 of 9 part products.
  fp16x3   candidate for a later round: two-part fp16 split (hi = fp16(x), lo = fp16(x - hi)),
           products hh + hl + lh, with power-of-two scales so that the parts stay inside fp16's exponent range:
           one scale for the weight matrix, one PER COLUMN of the activation operand (a column of the B tile is a lane of the
           MFMA operand and owns its accumulator column, so the scale can be undone per lane in the epilogue).
Part products are formed exactly (fp64 of exactly representable parts) and accumulated in fp32 like the MFMA does."""
import torch

torch.manual_seed(0)


def trunc_bf16(x):
    return (x.view(torch.int32) & -65536).view(torch.float32)


def bf16x6(A, B):
    def split(x):
        h = trunc_bf16(x); r = x - h; m = trunc_bf16(r); return h, m, trunc_bf16(r - m)
    ah, am, al = split(A); bh, bm, bl = split(B)
    return (ah @ bl + al @ bh) + am @ bm + (ah @ bm + am @ bh) + ah @ bh


def fp16x3(A, B, scale_b="column"):
    def split(x):
        h = x.half().float(); return h, (x - h).half().float()
    sa = 2.0 ** (8 - torch.floor(torch.log2(A.abs().max())))                 # weights: max -> [2^8, 2^9)
    if scale_b == "column":
        amax = B.abs().amax(0, keepdim=True).clamp_min(1e-30)
    else:
        amax = B.abs().max()
    sb = 2.0 ** (8 - torch.floor(torch.log2(amax)))
    ah, al = split(A * sa); bh, bl = split(B * sb)
    return (ah @ bl + al @ bh + ah @ bh) / (sa * sb)


def cases(K=512, M=256, N=512):
    A = torch.randn(M, K) * K ** -0.5
    return A, [("unit-variance activations", torch.randn(K, N)),
               ("channels spread over e^+-4", torch.randn(K, N) * torch.exp(4 * torch.randn(K, 1))),
               ("gradient-like, 1e-6 overall", 1e-6 * torch.randn(K, N) * torch.exp(2 * torch.randn(K, 1))),
               ("columns spread over e^+-6", torch.randn(K, N) * torch.exp(6 * torch.randn(1, N))),
               ("sparse spikes (1e4) in noise", torch.randn(K, N) + 1e4 * (torch.rand(K, N) < 1e-3))]


def errors(A, B):
    ref = A.double() @ B.double()
    den = A.double().abs() @ B.double().abs()                                # the natural scale of a dot product's rounding error
    out = {"fp32": A @ B, "bf16x6": bf16x6(A, B), "fp16x3 col-scale": fp16x3(A, B), "fp16x3 one scale": fp16x3(A, B, "tensor")}
    return {k: ((v.double() - ref).abs() / den).max().item() for k, v in out.items()}


if __name__ == "__main__":
    A, cs = cases()
    print("max |err| / (|A| |B|)  (fp32 rounding of a length-512 dot product is ~1e-7)")
    for name, B in cs:
        print("{:34s}".format(name) + "  ".join("{} {:.1e}".format(k, v) for k, v in errors(A, B).items()))
