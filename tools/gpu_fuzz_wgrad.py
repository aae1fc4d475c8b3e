"""Development tool: random shapes through the f16x3 weight-gradient kernel ON THE DEVICE against an fp64 product (the host simulation cannot
see asynchronous hazards of the raw rings: short slabs with 1 - 3 chunk pairs, slabs crossing sample boundaries, ragged frame counts).
    python tools/gpu_fuzz_wgrad.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-based_source_separation_amd", "src"))
import torch  # noqa: E402
import sepkernels  # noqa: E402
from sepkernels import PRO_GLN, PRO_GLN_PRELU, PRO_PRELU, STATS_SLOTS  # noqa: E402

K = sepkernels.HipBackend()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
worst = 0.0
for case in range(cases):
    B, M, N = ri(1, 5), 256 * ri(1, 2), 128 * ri(1, 3)
    T = ri(17, 700) if case % 3 else ri(2000, 4000)
    ldt = (T + 127) // 128 * 128
    chunks = B * (ldt // 32)
    ns = ri(1, min(48, chunks))
    mode = [0, PRO_PRELU, PRO_GLN, PRO_GLN_PRELU][ri(0, 3)]
    G = torch.zeros(B, M, ldt)
    X = torch.zeros(B, N, ldt)
    G[:, :, :T] = torch.randn(B, M, T, generator=g) * torch.exp(3 * torch.randn(B, M, 1, generator=g))
    X[:, :, :T] = torch.randn(B, N, T, generator=g) * torch.exp(2 * torch.randn(B, N, 1, generator=g))
    kw = {}
    Xe = X.double()
    if mode in (PRO_PRELU, PRO_GLN_PRELU):
        al = torch.tensor([0.25])
        kw.update(x_alpha=al.cuda())
        Xe = torch.where(Xe > 0, Xe, 0.25 * Xe)
    if mode in (PRO_GLN, PRO_GLN_PRELU):
        u = Xe[:, :, :T]
        cnt = float(N * T)
        st = torch.zeros(B, STATS_SLOTS, 2, dtype=torch.float64)
        st[:, 0, 0], st[:, 0, 1] = u.sum((1, 2)), (u * u).sum((1, 2))
        mu = st[:, 0, 0] / cnt
        rstd = 1.0 / torch.sqrt((st[:, 0, 1] / cnt - mu * mu).clamp_min(0) + 1e-8)
        gam, bet = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
        kw.update(x_stats=st.cuda(), x_gamma=gam.cuda(), x_beta=bet.cuda(), count=cnt)
        Xe = (Xe - mu.view(B, 1, 1)) * rstd.view(B, 1, 1) * gam.double().view(1, N, 1) + bet.double().view(1, N, 1)
        Xe[:, :, T:] = 0
    want = torch.einsum("bmt,bnt->mn", G.double()[:, :, :T], Xe[:, :, :T])
    wb = G.double()[:, :, :T].sum((0, 2))
    part, pb = torch.full((ns, M, N), float("nan")).cuda(), torch.full((ns, M), float("nan")).cuda()
    K.pw_wgrad(B=B, M=M, N=N, T=T, ldt=ldt, G=G.cuda(), X=X.cuda(), partial=part, partial_bias=pb, nsplit=ns, eps=1e-8, x_mode=mode, **kw)
    torch.cuda.synchronize()
    got, gb = part.double().sum(0).cpu(), pb.double().sum(0).cpu()
    # error model: relative to |G||X| per output (as for fp32 accumulation)
    bound = torch.einsum("bmt,bnt->mn", G.double()[:, :, :T].abs(), Xe[:, :, :T].abs())
    err = ((got - want).abs() / (bound + 1e-300)).max().item()
    eb = ((gb - wb).abs() / (G.double()[:, :, :T].abs().sum((0, 2)) + 1e-300)).max().item()
    worst = max(worst, err, eb)
    ok = torch.isfinite(got).all() and err < 3e-6 and eb < 3e-6
    if not ok or case % 10 == 0:
        print("case {:3d} B={} M={} N={} T={} ns={} mode={}  err {:.2e} bias {:.2e} {}".format(case, B, M, N, T, ns, mode, err, eb, "ok" if ok else "FAIL"), flush=True)
    if not ok:
        sys.exit(1)
print("{} cases, worst error relative to |G||X|: {:.2e}".format(cases, worst))
