"""GPU: every entry point of the C ABI (called through ctypes, i.e. through libsepkernels.so) against its CPU
emulation (tests/emulator.py) on identical seeded buffers.  The emulator itself is pinned to the oracle / the
reference golden vectors by the CPU tests, so agreement here chains each HIP kernel to the reference.
Pure outputs are pre-filled with NaN so that an element the kernel forgets to write is caught."""
import itertools

import pytest
import torch

import sepkernels
from sepkernels import (EPI_PRELU_BWD, EPI_RESIDUAL, EPI_ROWSUMS, EPI_ROWSUMS_PRELU, EPI_SIGMOID, EPI_STATS_PRELU, PRO_GLN,
                        PRO_GLN_BWD, PRO_GLN_PRELU, PRO_PRELU)
from emulator import EmuBackend

pytestmark = pytest.mark.gpu

EMU = EmuBackend()
HIP = sepkernels.HipBackend()
G = torch.Generator().manual_seed(1234)


PACKED = [False]


@pytest.fixture(params=["bf16x6", "f32", "f16x3", "f16x3-packed"])
def arith(request):
    """The arithmetics of sep_pw_gemm / sep_pw_wgrad (SEP_ARITH_*): exact three-way bf16 split on the bf16 MFMA, the fp32
    MFMA, the scaled two-part fp16 split (stand-alone calls form the |A| bound themselves), and the same with the weights
    split beforehand by sep_pack_weights (sep_gemm_desc.A_pk: the form the Conv-TasNet step uses; shapes the packed kernel
    does not take fall through to the fp32 weights inside sep_pw_gemm)."""
    prev = sepkernels.set_gemm_arith(request.param.split("-")[0])
    PACKED[0] = request.param.endswith("packed")
    yield request.param
    PACKED[0] = False
    sepkernels.set_gemm_arith(prev)


def packed_operand(gkw):
    """A_pk of a pw_gemm call from its device-side A / A2 / trans_a, as net.pack_weights builds it."""
    A, A2, M, K = gkw["A"], gkw.get("A2"), gkw["M"], gkw["K"]
    if gkw.get("trans_a"):
        W = A.reshape(-1, M) if A2 is None else torch.cat([A.reshape(-1, M), A2.reshape(-1, M)], 0).contiguous()
        return HIP.pack_weights([(W, K, M, 1)])[0]
    assert A2 is None
    return HIP.pack_weights([(A.reshape(M, K), M, K, 0)])[0]


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=G) * scale).float()


def nan(*shape):
    return torch.full(shape, float("nan"))


def padded(B, C, T, ldt, scale=1.0):
    x = rnd(B, C, ldt, scale=scale)
    x[..., T:] = 0
    return x


SLOTS = sepkernels.STATS_SLOTS


def zstats(B):
    return torch.zeros(B, SLOTS, 2, dtype=torch.float64)


def stats_of(x, T):
    """slotted statistics tensor with the totals of x[..., :T] spread unevenly over the slots"""
    v = x[..., :T].double()
    tot = torch.stack([v.sum((1, 2)), (v * v).sum((1, 2))], 1)      # (B, 2)
    w = torch.rand(SLOTS, generator=G).double()
    w = w / w.sum()
    return (tot.unsqueeze(1) * w.view(1, SLOTS, 1)).contiguous()


def bacc_rand(B, count, scale=0.01):
    """slotted gLN-backward sums (sep_gemm_desc.pro_bacc): random raw totals {sum gamma g, sum gamma g u} of the size `scale * count`,
    spread unevenly over the slots"""
    tot = (torch.randn(B, 2, generator=G) * scale * count).double()
    w = torch.rand(SLOTS, generator=G).double()
    w = w / w.sum()
    return (tot.unsqueeze(1) * w.view(1, SLOTS, 1)).contiguous()


def to_device(t):
    """where the second copy of every buffer lives (tests/test_kernel_source_on_host_cpu.py swaps HIP, to_device and device_sync to run
    these same cases on the host simulation of the kernel sources)"""
    return t.cuda()


def device_sync():
    torch.cuda.synchronize()


def device_name():
    return "cuda"


def both(op, args, kwargs=None, tol=2e-4, names=None):
    """Run `op` on the emulator (CPU tensors) and on HIP (cuda clones); compare every tensor argument afterwards."""
    kwargs = kwargs or {}
    memo = {}

    def to_gpu(v):   # aliasing arguments (in-place ops) must stay aliased on the device
        if not torch.is_tensor(v):
            return v
        if id(v) not in memo:
            memo[id(v)] = to_device(v)
        return memo[id(v)]
    gargs = [to_gpu(a) for a in args]
    gkw = {k: to_gpu(v) for k, v in kwargs.items()}
    getattr(EMU, op)(*args, **kwargs)
    if op == "pw_gemm" and PACKED[0] and gkw["K"] % 16 == 0:
        gkw["A_pk"] = packed_operand(gkw)
    getattr(HIP, op)(*gargs, **gkw)
    device_sync()
    items = list(enumerate(zip(args, gargs))) + [(k, (kwargs[k], gkw[k])) for k in kwargs]      # (A_pk exists on one side only)
    for key, (c, g) in items:
        if not torch.is_tensor(c):
            continue
        gc = g.cpu()
        if c.dtype == torch.float64 and c.dim() >= 2 and tuple(c.shape[-2:]) == (SLOTS, 2):
            c, gc = c.sum(-2), gc.sum(-2)       # slot distribution is an implementation detail: compare totals
        if c.dtype in (torch.int64, torch.int32):
            assert torch.equal(c, gc), "{}: integer output {} differs".format(op, key)
            continue
        assert torch.isfinite(gc).all() == torch.isfinite(c).all(), "{}: arg {} finite-ness differs (unwritten output?)".format(op, key)
        assert torch.isfinite(gc).all(), "{}: arg {} has non-finite values".format(op, key)
        ref = c.double()
        err = (gc.double() - ref).abs().max().item()
        den = ref.abs().max().item() + 1e-30
        assert err <= tol * den, "{}: arg {} max err {:.3e} vs scale {:.3e}".format(op, key, err, den)


# ------------------------------------------------------------------------------------------- encoder / unfold
@pytest.mark.parametrize("Cin,L,S,relu,pad_left", [(1, 16, 8, 0, 0), (1, 16, 8, 1, 3), (2, 20, 10, 1, 4), (1, 2, 1, 0, 0)])
def test_encoder_and_unfold(Cin, L, S, relu, pad_left):
    B, N, Tin = 2, 64, 1999
    F = (Tin + 2 * pad_left - L) // S + 1
    ldt = (F + 127) // 128 * 128
    x, E = rnd(B, Cin, Tin), rnd(N, Cin, L)
    both("encoder_fwd", [x, E, nan(B, N, ldt), zstats(B), B, Cin, Tin, N, L, S, F, ldt, pad_left, relu])
    both("unfold", [x, nan(B, Cin * L, ldt), B, Cin, Tin, L, S, F, ldt, pad_left])


# ------------------------------------------------------------------------------------------- pointwise GEMM
def _gemm_common(B, M, K, T):
    ldt = (T + 127) // 128 * 128
    return ldt, padded(B, K, T, ldt), rnd(M, K, scale=K ** -0.5), rnd(M)


@pytest.mark.parametrize("B,M,K,T", [(2, 64, 128, 300), (1, 512, 128, 3999), (3, 128, 64, 129)])
def test_gemm_plain_bias(B, M, K, T, arith):
    ldt, X, A, bias = _gemm_common(B, M, K, T)
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=nan(B, M, ldt), bias=bias))


@pytest.mark.parametrize("M,K,T", [(16, 32, 201), (64, 32, 201), (32, 16, 130), (48, 48, 77)])
def test_gemm_small_widths_of_the_dual_path_separators(M, K, T, arith):
    """the 1x1 convolutions of the small DPTNet / GALRNet / SepFormer fixtures (tests/golden/{dptnet,galrnet,sepformer}*.npz):
    fewer rows than one tile, contraction lengths of one to three ring stages -- plain, with the PReLU prologue, and both
    input-gradient forms (plain, PReLU-derivative epilogue)"""
    if PACKED[0]:
        pytest.skip("stand-alone callers (the only users of these widths) hand over fp32 weights; sep_pack_weights needs M % 32 == 0")
    B = 2
    ldt, X, A, bias = _gemm_common(B, M, K, T)
    alpha = torch.tensor([0.3])
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=nan(B, M, ldt), bias=bias))
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=nan(B, M, ldt), bias=bias, pro_mode=PRO_PRELU, pro_alpha=alpha))
    dY = padded(B, M, T, ldt)
    both("pw_gemm", [], dict(B=B, M=K, K=M, T=T, ldt=ldt, trans_a=1, A=A, X=dY, Y=nan(B, K, ldt)))
    both("pw_gemm", [], dict(B=B, M=K, K=M, T=T, ldt=ldt, trans_a=1, A=A, X=dY, Y=nan(B, K, ldt), epi_flags=EPI_PRELU_BWD, epi_aux=X,
                             epi_alpha=alpha, epi_dalpha=torch.zeros(1, dtype=torch.float64)))


def test_gemm_gln_prologue_and_stats_epilogue(arith):
    B, M, K, T = 2, 128, 256, 777
    ldt, X, A, bias = _gemm_common(B, M, K, T)
    X = X * 2 + 0.3
    X[..., T:] = 0
    st = stats_of(X, T)
    gamma, beta, alpha = rnd(K) + 1, rnd(K), torch.tensor([0.2])
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=nan(B, M, ldt), bias=bias, pro_mode=PRO_GLN, pro_stats=st,
                             pro_gamma=gamma, pro_beta=beta, count=K * T, eps=1e-12, epi_flags=EPI_STATS_PRELU,
                             epi_alpha=alpha, epi_stats=zstats(B)))


@pytest.mark.parametrize("B,M,T", [(2, 256, 333), (3, 512, 1000), (1, 128, 4096)])
def test_gemm_conv1_shape_k128(B, M, T, arith):
    """The TCN conv1 form at its real contraction length (K = 128): also the shape of the persistent variant (SEPK_PERSIST=1)."""
    K = 128
    ldt, X, A, bias = _gemm_common(B, M, K, T)
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=nan(B, M, ldt), bias=bias, epi_flags=EPI_STATS_PRELU,
                             epi_alpha=torch.tensor([0.25]), epi_stats=zstats(B), eps=1e-12))


@pytest.mark.parametrize("M,T", [(256, 333), (128, 128), (192, 200)])
def test_gemm_stats_epilogue_specialised_and_generic(M, T, arith):
    """TCN conv1 form (no prologue, PReLU statistics): M % 128 == 0 takes the compile-time-flag instantiation without row
    predicates, M = 192 the run-time-flag one; T off the 128 grid exercises the column pre-mask of the edge tile."""
    B, K = 2, 64
    ldt, X, A, bias = _gemm_common(B, M, K, T)
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=nan(B, M, ldt), bias=bias, epi_flags=EPI_STATS_PRELU,
                             epi_alpha=torch.tensor([0.25]), epi_stats=zstats(B), eps=1e-12))


def test_gemm_packed_heads_residual_accumulate(arith):
    """[Wo;Ws] as one operand: rows < m_split -> Y (+residual), rows >= m_split -> Y2 (+=)."""
    B, Bn, Sc, H, T = 2, 128, 128, 256, 500
    ldt = 512
    z = padded(B, H, T, ldt) * 1.5 + 0.2
    z[..., T:] = 0
    u = torch.where(z > 0, z, 0.3 * z)
    st = stats_of(u, T)
    A, bias = rnd(Bn + Sc, H, scale=H ** -0.5), rnd(Bn + Sc)
    res, skip0 = padded(B, Bn, T, ldt), padded(B, Sc, T, ldt)
    kw = dict(B=B, M=Bn + Sc, K=H, T=T, ldt=ldt, A=A, X=z, Y=nan(B, Bn, ldt), Y2=skip0, m_split=Bn, bias=bias, accumulate=1,
              epi_flags=EPI_RESIDUAL, epi_res=res, pro_mode=PRO_GLN_PRELU, pro_stats=st, pro_gamma=rnd(H) + 1, pro_beta=rnd(H),
              pro_alpha=torch.tensor([0.3]), count=H * T, eps=1e-12)
    both("pw_gemm", [], kw)


def test_gemm_heads_plain_input_residual_accumulate(arith):
    """the same joint product on an input that is already normalised (the staged causal layers: cLN is a pass of its own): no prologue,
    residual on the first m_split rows, accumulation on the rest"""
    B, Bn, Sc, H, T = 2, 128, 128, 256, 500
    ldt = 512
    v = padded(B, H, T, ldt) * 1.5 + 0.2
    A, bias = rnd(Bn + Sc, H, scale=H ** -0.5), rnd(Bn + Sc)
    kw = dict(B=B, M=Bn + Sc, K=H, T=T, ldt=ldt, A=A, X=v, Y=nan(B, Bn, ldt), Y2=padded(B, Sc, T, ldt), m_split=Bn, bias=bias, accumulate=1,
              epi_flags=EPI_RESIDUAL, epi_res=padded(B, Bn, T, ldt))
    both("pw_gemm", [], kw)


def test_gemm_prelu_prologue_sigmoid(arith):
    B, M, K, T = 2, 384, 64, 260
    ldt, X, A, bias = _gemm_common(B, M, K, T)
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=A, X=X, Y=nan(B, M, ldt), bias=bias, pro_mode=PRO_PRELU,
                             pro_alpha=torch.tensor([0.25]), epi_flags=EPI_SIGMOID))


def test_gemm_dgrad_two_sources_rowsums(arith):
    """dv2 = Wo^T dout + Ws^T dS with the gLN-backward row sums in the epilogue."""
    B, Bn, Sc, H, T = 2, 128, 64, 256, 700
    ldt = 768
    Wo, Ws = rnd(Bn, H, scale=0.1), rnd(Sc, H, scale=0.1)
    dout, dS, z = padded(B, Bn, T, ldt), padded(B, Sc, T, ldt), padded(B, H, T, ldt)
    kw = dict(B=B, M=H, K=Bn + Sc, T=T, ldt=ldt, trans_a=1, A=Wo, A2=Ws, X=dout, X2=dS, k_split=Bn, Y=nan(B, H, ldt),
              epi_flags=EPI_ROWSUMS | EPI_ROWSUMS_PRELU, epi_aux=z, epi_alpha=torch.tensor([0.15]),
              epi_rowpart=nan(B, H, ldt // 64, 2))
    both("pw_gemm", [], kw)


@pytest.mark.parametrize("H,T", [(512, 700), (256, 130), (512, 1000)])      # the last: heads^T on the persistent producer / consumer kernel (packed weights)
def test_gemm_dgrad_two_sources_plain(H, T, arith):
    """dv2 = Wo^T dout + Ws^T dS with a plain epilogue: the heads^T product of the step since its gLN sums come from the weight gradient
    (H = 512: the 128-row-per-wave form of the cooperative kernel when SEPK_COOP_MI=4)"""
    B, Bn, Sc = 2, 128, 128
    ldt = (T + 127) // 128 * 128
    Wo, Ws = rnd(Bn, H, scale=0.1), rnd(Sc, H, scale=0.1)
    kw = dict(B=B, M=H, K=Bn + Sc, T=T, ldt=ldt, trans_a=1, A=Wo, A2=Ws, X=padded(B, Bn, T, ldt), X2=padded(B, Sc, T, ldt), k_split=Bn, Y=nan(B, H, ldt))
    both("pw_gemm", [], kw)


def test_gemm_dgrad_prelu_bwd(arith):
    B, M, K, T = 2, 64, 384, 333
    ldt = 384
    Wm = rnd(K, M, scale=0.1)
    dpre, S = padded(B, K, T, ldt), padded(B, M, T, ldt)
    both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, trans_a=1, A=Wm, X=dpre, Y=nan(B, M, ldt), epi_flags=EPI_PRELU_BWD,
                             epi_aux=S, epi_alpha=torch.tensor([0.25]), epi_dalpha=torch.zeros(1, dtype=torch.float64)))


@pytest.mark.parametrize("residual", [0, 1])
def test_gemm_gln_bwd_prologue(residual, arith):
    B, M, K, T = 2, 128, 256, 450
    ldt = 512
    W1 = rnd(K, M, scale=0.1)
    a = padded(B, K, T, ldt)
    u = torch.where(a > 0, a, 0.2 * a)
    st = stats_of(u, T)
    dv = padded(B, K, T, ldt)
    kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, trans_a=1, A=W1, X=dv, Y=nan(B, M, ldt), pro_mode=PRO_GLN_BWD, pro_stats=st,
              pro_gamma=rnd(K) + 1, pro_alpha=torch.tensor([0.2]), pro_aux=a, pro_bsum=rnd(B, 2, scale=0.01), pro_store=dv,
              pro_dalpha=torch.zeros(1, dtype=torch.float64), count=K * T, eps=1e-12)
    if residual:      # ... and the means formed by the kernel from a producer's slots (pro_bacc) instead of published floats
        kw.update(epi_flags=EPI_RESIDUAL, epi_res=padded(B, M, T, ldt), pro_bsum=None, pro_bacc=bacc_rand(B, K * T))
    both("pw_gemm", [], kw)


@pytest.mark.parametrize("M,K", [(256, 64), (192, 48)])
def test_gemm_gln_bwd_prologue_several_row_tiles(M, K, arith):
    """more than one 128-row tile: da goes to its own buffer (row tile 0 stores it, every tile recomputes it from the untouched X),
    and handing X over as pro_store is refused -- by the library and by the restatement alike"""
    B, T, ldt = 2, 300, 512
    a = padded(B, K, T, ldt)
    dv = padded(B, K, T, ldt)
    kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, trans_a=1, A=rnd(K, M, scale=0.1), X=dv, Y=nan(B, M, ldt), pro_mode=PRO_GLN_BWD,
              pro_stats=stats_of(torch.where(a > 0, a, 0.2 * a), T), pro_gamma=rnd(K) + 1, pro_alpha=torch.tensor([0.2]), pro_aux=a,
              pro_bsum=rnd(B, 2, scale=0.01), pro_store=nan(B, K, ldt), pro_dalpha=torch.full((1,), 100.0, dtype=torch.float64), count=K * T, eps=1e-12,
              epi_flags=EPI_RESIDUAL, epi_res=padded(B, M, T, ldt))
    both("pw_gemm", [], kw)
    kw["pro_store"] = kw["X"]
    with pytest.raises(RuntimeError, match="one row tile"):
        both("pw_gemm", [], kw)
    dev = to_device(dv)
    with pytest.raises(RuntimeError, match="one row tile"):
        HIP.pw_gemm(**{k: (dev if v is dv else to_device(v) if torch.is_tensor(v) else v) for k, v in kw.items()})


@pytest.mark.parametrize("alpha", [-0.3, 0.0, 1.0, 1.7])
def test_gemm_prelu_prologues_any_slope(alpha):
    """The producer / consumer kernel has a two-instruction PReLU (max(x, alpha x)) for 0 <= alpha <= 1 and the general form
    otherwise: both, at the boundaries, for the three prologues that contain a PReLU (packed weights, K multiple of 64)."""
    PACKED[0] = True
    try:
        al = torch.tensor([alpha])
        B, Bn, Sc, H, T, ldt = 2, 128, 128, 512, 500, 512           # K = 512, M = 1024, K = 512: the shapes the dispatch gives to gemm_pc.hip
        z = padded(B, H, T, ldt) * 1.5 + 0.2
        z[..., T:] = 0
        st = stats_of(torch.where(z > 0, z, alpha * z), T)
        both("pw_gemm", [], dict(B=B, M=Bn + Sc, K=H, T=T, ldt=ldt, A=rnd(Bn + Sc, H, scale=H ** -0.5), X=z, Y=nan(B, Bn, ldt), Y2=padded(B, Sc, T, ldt),
                                 m_split=Bn, bias=rnd(Bn + Sc), accumulate=1, epi_flags=EPI_RESIDUAL, epi_res=padded(B, Bn, T, ldt), pro_mode=PRO_GLN_PRELU,
                                 pro_stats=st, pro_gamma=rnd(H) + 1, pro_beta=rnd(H), pro_alpha=al, count=H * T, eps=1e-12))
        M, K = 1024, 128
        both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, A=rnd(M, K, scale=K ** -0.5), X=padded(B, K, T, ldt), Y=nan(B, M, ldt), bias=rnd(M),
                                 pro_mode=PRO_PRELU, pro_alpha=al, epi_flags=EPI_SIGMOID))
        M, K = 128, 512
        a = padded(B, K, T, ldt)
        dv = padded(B, K, T, ldt)
        both("pw_gemm", [], dict(B=B, M=M, K=K, T=T, ldt=ldt, trans_a=1, A=rnd(K, M, scale=0.1), X=dv, Y=nan(B, M, ldt), pro_mode=PRO_GLN_BWD,
                                 pro_stats=stats_of(torch.where(a > 0, a, alpha * a), T), pro_gamma=rnd(K) + 1, pro_alpha=al, pro_aux=a,
                                 pro_bsum=rnd(B, 2, scale=0.01), pro_store=dv, pro_dalpha=torch.zeros(1, dtype=torch.float64), count=K * T, eps=1e-12,
                                 epi_flags=EPI_RESIDUAL, epi_res=padded(B, M, T, ldt)))
    finally:
        PACKED[0] = False


@pytest.mark.parametrize("M,K,T", [(128, 512, 3999), (256, 512, 1000), (512, 128, 3999), (1024, 128, 700), (128, 1024, 450), (512, 256, 2100)])
def test_gemm_packed_weights_model_shapes(M, K, T):
    """The packed-weight kernel at the (M, K) pairs of the paper-best model -- one and two 32-row blocks per wave, short and
    long contractions, edge / dead column tiles -- against fp64, beside the fp32-MFMA kernel on the same operands; weights
    with rows of very different magnitude (the packer scales per row) and X rows spread over e^+-3 (per-column scales)."""
    B = 2
    ldt = (T + 127) // 128 * 128
    X = padded(B, K, T, ldt) * torch.exp(3 * rnd(B, K, 1))
    X[..., T:] = 0
    A = rnd(M, K, scale=K ** -0.5) * torch.exp(4 * rnd(M, 1))
    bias = rnd(M)
    ref = torch.einsum("mk,bkt->bmt", A.double(), X.double()) + bias.double().view(1, M, 1)
    scale = torch.einsum("mk,bkt->bmt", A.double().abs(), X.double().abs())[..., :T] + bias.double().abs().view(1, M, 1) + 1e-30
    err = {}
    for name in ("f32", "packed"):
        Y = torch.full((B, M, ldt), float("nan"), device=device_name())
        Ag = to_device(A)
        kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, A=Ag, X=to_device(X), Y=Y, bias=to_device(bias))
        if name == "packed":
            kw.update(arith=sepkernels.ARITH_F16X3, A_pk=HIP.pack_weights([(Ag, M, K, 0)])[0])
        else:
            kw.update(arith=sepkernels.ARITH_F32)
        HIP.pw_gemm(**kw)
        device_sync()
        assert torch.isfinite(Y).all()
        assert (Y[..., T:] == 0).all()
        err[name] = (((Y.cpu().double() - ref)[..., :T]).abs() / scale).max().item()
    assert err["f32"] <= 5e-6, err
    assert err["packed"] <= max(3 * err["f32"], 6e-7), err


@pytest.mark.parametrize("case", ["row_scales_1e12", "subnormal_activations", "late_jump_k1024", "zero_rows_and_columns"])
def test_gemm_packed_adversarial_operands(case):
    """SEP_ARITH_F16X3 (packed weights) where a shared scale would fail: weight rows 24 decades apart; activations in fp32's
    subnormal range; a column whose magnitude jumps by 1e6 in the LAST chunk of a K = 1024 contraction (accumulator rescale path);
    all-zero rows of A and columns of X.  Error relative to |A||X| per output, beside the fp32-MFMA kernel on the same operands."""
    B, M, K, T = 1, 256, 1024 if case == "late_jump_k1024" else 512, 700
    ldt = 768
    X = padded(B, K, T, ldt)
    A = rnd(M, K, scale=K ** -0.5)
    if case == "row_scales_1e12":
        A[::2] *= 1e12
        A[1::2] *= 1e-12
    elif case == "subnormal_activations":
        X = X * 1e-39                                    # |x| ~ 1e-39 < 1.18e-38 = the smallest normal fp32
    elif case == "late_jump_k1024":
        X[:, K - 16:, ::3] *= 1e6
    else:
        A[5] = 0
        A[77] = 0
        X[:, :, 10:200:7] = 0
    X[..., T:] = 0
    ref = torch.einsum("mk,bkt->bmt", A.double(), X.double())
    scale = torch.einsum("mk,bkt->bmt", A.double().abs(), X.double().abs())[..., :T]
    floor = 1.4e-45 * A.double().abs().sum(1).view(1, M, 1)          # one fp32 ulp at the bottom of the subnormal range per term
    err = {}
    for name in ("f32", "packed"):
        Y = torch.full((B, M, ldt), float("nan"), device=device_name())
        Ag = to_device(A)
        kw = dict(B=B, M=M, K=K, T=T, ldt=ldt, A=Ag, X=to_device(X), Y=Y)
        kw.update(arith=sepkernels.ARITH_F16X3, A_pk=HIP.pack_weights([(Ag, M, K, 0)])[0]) if name == "packed" else kw.update(arith=sepkernels.ARITH_F32)
        HIP.pw_gemm(**kw)
        device_sync()
        assert torch.isfinite(Y).all()
        e = ((Y.cpu().double() - ref)[..., :T]).abs()
        err[name] = (e / (scale + floor + 1e-300)).max().item()
        if case == "zero_rows_and_columns":
            assert (Y[:, 5, :T] == 0).all() and (Y[:, :, 10:200:7] == 0).all()
    assert err["f32"] <= 1e-5, err
    assert err["packed"] <= max(3 * err["f32"], 2e-6), err


@pytest.mark.parametrize("case", ["row_scales_1e12", "late_jumps", "zero_rows_then_signal", "subnormal", "one_slab_many_samples"])
def test_wgrad_f16_adversarial_operands(case):
    """sep_pw_wgrad in SEP_ARITH_F16X3 (wgrad_pc16.hip: both operands scaled per row at run time) where fixed scales would fail: rows of G
    and of X 24 decades apart; rows whose magnitude jumps by 1e6 late in a slab (the accumulator-row and accumulator-column rescale
    paths, both at once); rows that are all zero for the first half of the slab; operands in fp32's subnormal range; one slab running
    over several samples.  Error of the summed slabs relative to sum_t |G||X| per output, beside the exact bf16x6 kernel on the same operands."""
    B, M, N, T, ns = 3, 256, 128, 1900, (1 if case == "one_slab_many_samples" else 5)
    ldt = 1920
    Gm, X = padded(B, M, T, ldt), padded(B, N, T, ldt)
    if case == "row_scales_1e12":
        Gm[:, ::2] *= 1e12
        Gm[:, 1::2] *= 1e-12
        X[:, ::3] *= 1e9
        X[:, 1::3] *= 1e-9
    elif case == "late_jumps":
        Gm[:, 3::7, 1500:] *= 1e6
        X[:, 5::11, 1700:] *= 1e6
        Gm[1, 40, 1899] = 3e4
    elif case == "zero_rows_then_signal":
        Gm[:, ::4, :1000] = 0
        X[:, ::5, :1200] = 0
        Gm[:, 9] = 0
        X[:, 17] = 0
    elif case == "subnormal":
        Gm, X = Gm * 1e-20, X * 1e-19                        # products ~1e-39: fp32 subnormals in the accumulators
    Gm[..., T:] = 0
    X[..., T:] = 0
    ref = torch.einsum("bmt,bnt->mn", Gm.double(), X.double())
    scale = torch.einsum("bmt,bnt->mn", Gm.double().abs(), X.double().abs())
    err = {}
    for name, arith in (("bf16x6", sepkernels.ARITH_BF16X6), ("f16x3", sepkernels.ARITH_F16X3)):
        part = torch.full((ns, M, N), float("nan"), device=device_name())
        pb = torch.full((ns, M), float("nan"), device=device_name())
        HIP.pw_wgrad(B=B, M=M, N=N, T=T, ldt=ldt, G=to_device(Gm), X=to_device(X), partial=part, partial_bias=pb, nsplit=ns, arith=arith)
        device_sync()
        assert torch.isfinite(part).all() and torch.isfinite(pb).all()
        got = part.cpu().double().sum(0)
        err[name] = ((got - ref).abs() / (scale + 1e-300 + 1.4e-45 * T)).max().item()
        assert (pb.cpu().double().sum(0) - Gm.double().sum((0, 2))).abs().max() <= 1e-5 * Gm.double().abs().sum((0, 2)).max()
        if case == "zero_rows_then_signal":
            assert (got[9] == 0).all() and (got[:, 17] == 0).all()
    assert err["bf16x6"] <= 2e-6, err
    assert err["f16x3"] <= max(4 * err["bf16x6"], 2e-6), err


def test_pack_weights_reproduces_the_weights():
    """hi + lo of every packed group, times the row's inverse scale, is the weight to 2^-22 relative to the row maximum."""
    W = rnd(96, 64) * torch.exp(5 * rnd(96, 1))
    W[5] = 0
    for trans in (0, 1):
        pk = HIP.pack_weights([(to_device(W), 96, 64, trans)])[0]
        device_sync()
        M, K = pk.M, pk.K
        # operand-block layout: [m / 32][k / 16][hi | lo][lane = 32 * ((k >> 3) & 1) + (m & 31)][8 fp16]
        h = pk.data.view(torch.float16).view(M // 32, K // 16, 2, 2, 32, 8).float().cpu()
        back = (h[:, :, 0] + h[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(M, K).double() * pk.rscale.cpu().double().view(M, 1)
        A = (W.t() if trans else W).double()
        assert (back - A).abs().max() <= 2.0 ** -21 * A.abs().amax(1, keepdim=True).clamp_min(1e-30).max()
        assert ((back - A).abs() <= 2.0 ** -21 * A.abs().amax(1, keepdim=True)).all()


# ------------------------------------------------------------------------------------------- weight gradient
def _wg_out(ns, M, N):
    return nan(ns, M, N), nan(ns, M)


def _reduce_check(part_c, part_g, tol=3e-4):
    a, b = part_c.double().sum(0), part_g.double().cpu().sum(0)
    assert (a - b).abs().max() <= tol * (a.abs().max() + 1e-30)


def _wgrad_both(kw, tol=3e-4):
    """Slab contents differ by construction (the emulator puts everything in slab 0): compare the slab sums."""
    ck = dict(kw)
    gk = {k: (to_device(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    EMU.pw_wgrad(**ck)
    HIP.pw_wgrad(**gk)
    device_sync()
    assert torch.isfinite(gk["partial"]).all()
    _reduce_check(ck["partial"], gk["partial"], tol)
    if kw.get("partial_bias") is not None:
        assert torch.isfinite(gk["partial_bias"]).all()
        _reduce_check(ck["partial_bias"], gk["partial_bias"], tol)
    B, ns, ldt = kw["B"], kw["nsplit"], kw["ldt"]
    if ns % B == 0 and (ldt // 32) % (ns // B) == 0:
        # sample-aligned slabs (sepkernels.h): slab s holds frames of sample s // k only -- compare sample by sample
        k = ns // B
        pc, pg = ck["partial"].reshape(B, k, -1).sum(1).double(), gk["partial"].cpu().reshape(B, k, -1).sum(1).double()
        assert (pc - pg).abs().max() <= tol * pc.abs().max() + 1e-30, "per-sample slab sums differ"
        if kw.get("partial_bias") is not None:
            bc, bg = ck["partial_bias"].reshape(B, k, -1).sum(1).double(), gk["partial_bias"].cpu().reshape(B, k, -1).sum(1).double()
            assert (bc - bg).abs().max() <= tol * bc.abs().max() + 1e-30, "per-sample bias slab sums differ"


@pytest.mark.parametrize("K,scale", [(128, 1.0), (512, 1.0), (1024, 1e-3), (512, 1e4)])
def test_gemm_split_arithmetic_is_as_accurate_as_fp32_mfma(K, scale):
    """Against fp64: the errors of the bf16x6 and f16x3 paths are at the level of the fp32-MFMA path's (all are fp32
    products with fp32 accumulation), over wide dynamic range of the operands (the bf16 split is exact for any finite fp32
    value; the fp16 split is scaled per column of X).  The two-part fp16 split carries 22 significand bits against fp32's 24, i.e.
    up to 4x the rounding error of a single product; at short contractions that is what shows (K = 128: ratio 2.0 median, 3.6
    worst over 12 operand draws, tools/split_ratio.py on the GPU box), at K >= 512 the fp32 accumulation both share dominates (1.1-1.5)."""
    G.manual_seed(1000 + K)                                          # own operand draw: independent of which tests ran before
    B, M, T = 2, 256, 1000
    ldt = 1024
    X = padded(B, K, T, ldt) * torch.exp(3 * rnd(B, K, 1))          # rows of very different magnitude
    X[..., T:] = 0
    A = rnd(M, K) * scale
    ref = torch.einsum("mk,bkt->bmt", A.double(), X.double())
    err = {}
    for name in ("f32", "bf16x6", "f16x3"):
        Y = torch.full((B, M, ldt), float("nan"), device=device_name())
        HIP.pw_gemm(B=B, M=M, K=K, T=T, ldt=ldt, A=to_device(A), X=to_device(X), Y=Y, arith=sepkernels.arith_code(name))
        device_sync()
        d = (Y.cpu().double() - ref)[..., :T]
        err[name] = (d.abs().max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    assert err["bf16x6"][0] <= max(2 * err["f32"][0], 2e-7), err
    assert err["bf16x6"][1] <= max(2 * err["f32"][1], 1e-7), err
    assert err["f16x3"][0] <= max(4 * err["f32"][0], 3e-7), err          # error relative to |A||X| per output, like fp32's
    assert err["f16x3"][1] <= max(4 * err["f32"][1], 2e-7), err
    assert err["f32"][0] <= 5e-6, err


@pytest.mark.parametrize("B,M,N,T,ns", [(2, 256, 128, 999, 7), (1, 128, 16, 300, 3), (3, 64, 64, 130, 1), (2, 512, 128, 3999, 64),
                                        (2, 32, 4, 201, 2), (2, 16, 32, 201, 3), (4, 64, 32, 201, 1)])
def test_wgrad_plain(B, M, N, T, ns, arith):
    ldt = (T + 127) // 128 * 128
    part, pb = _wg_out(ns, M, N)
    _wgrad_both(dict(B=B, M=M, N=N, T=T, ldt=ldt, G=padded(B, M, T, ldt), X=padded(B, N, T, ldt), partial=part, partial_bias=pb, nsplit=ns))


@pytest.mark.parametrize("B,M,N,T,ns,n", [(2, 512, 128, 999, 5, 3), (1, 256, 128, 300, 2, 8), (2, 64, 32, 201, 2, 2)])
def test_wgrad_batch_equals_separate_calls(B, M, N, T, ns, n, arith):
    """sep_pw_wgrad_batch: n weight gradients of one shape in ONE launch (the conv1 weight gradients of consecutive TCN layers) give, slab
    by slab and bit for bit, what n calls of sep_pw_wgrad with the same nsplit give -- the batched grid only re-indexes the workgroups
    (shapes outside the fp16 producer / consumer kernel, and the other arithmetics, take the n calls inside the entry point)."""
    ldt = (T + 127) // 128 * 128
    ops = [(to_device(padded(B, M, T, ldt)), to_device(padded(B, N, T, ldt))) for _ in range(n)]
    single, batch = [], []
    for g, x in ops:
        part, pb = to_device(nan(ns, M, N)), to_device(nan(ns, M))
        HIP.pw_wgrad(B=B, M=M, N=N, T=T, ldt=ldt, G=g, X=x, partial=part, partial_bias=pb, nsplit=ns)
        single.append((part, pb))
    calls = []
    for g, x in ops:
        part, pb = to_device(nan(ns, M, N)), to_device(nan(ns, M))
        calls.append(dict(B=B, M=M, N=N, T=T, ldt=ldt, G=g, X=x, partial=part, partial_bias=pb, nsplit=ns))
        batch.append((part, pb))
    HIP.pw_wgrad_batch(calls)
    device_sync()
    for (p0, b0), (p1, b1), (g, x) in zip(single, batch, ops):
        assert torch.equal(p0.cpu(), p1.cpu()) and torch.equal(b0.cpu(), b1.cpu())
        ref = torch.einsum("bmt,bnt->mn", g.cpu().double(), x.cpu().double())
        assert (p1.cpu().double().sum(0) - ref).abs().max() <= 1e-4 * ref.abs().max()
    with pytest.raises(sepkernels.SepKernelsError, match="must agree"):
        HIP.pw_wgrad_batch([calls[0], dict(calls[1], nsplit=ns + 1)])


@pytest.mark.parametrize("B,N,T,k", [(2, 128, 999, 4), (1, 512, 3999, 4), (3, 128, 300, 1), (1, 128, 3999, 32)])
def test_wgrad_with_a_presplit_second_source(B, N, T, k):
    """sep_split_rows + sep_wgrad_desc.G2_pre (round 6): the heads' weight gradient with its second G source -- the skip gradient dS, the same
    tensor in all 24 layers -- handed over as {hi, lo} fp16 operand lines with one scale per (sample, row), split ONCE.  The split is exact to
    2^-22 of the row's maximum, its per-slab row sums are the bias partials, and the weight gradient equals the fp64 product as closely as the
    form that splits dS inside the kernel does (both against |G||X| per output)."""
    prev = sepkernels.set_gemm_arith("f16x3")
    try:
        M, gs = 256, 128
        ldt = (T + 127) // 128 * 128
        while (ldt // 32) % k:
            ldt += 128
        dout, dS, z = padded(B, gs, T, ldt), padded(B, M - gs, T, ldt, scale=3.0), padded(B, N, T, ldt)
        dS[:, 5] *= 1e-6                                    # rows of very different scale; a silent row
        dS[:, 7] = 0
        al = torch.tensor([0.2])
        gdout, gdS, gz, gal = (to_device(t) for t in (dout, dS, z, al))
        planes, exps, sums = HIP.split_rows(gdS, T, k)
        device_sync()
        # the planes decode to dS * 2^e: per 32-frame line 32 hi then 32 lo fp16
        raw = planes.cpu().contiguous().view(torch.float16).view(B, M - gs, ldt // 32, 2, 32).float()
        rec = (raw[:, :, :, 0] + raw[:, :, :, 1]).reshape(B, M - gs, ldt).double() * torch.pow(2.0, -exps.cpu().double()).unsqueeze(-1)
        rowmax = dS[..., :T].abs().amax(-1, keepdim=True).double()
        assert ((rec - dS.double()).abs() <= 2.0 ** -21 * rowmax + 1e-30).all()
        assert (rec[..., T:] == 0).all() and exps.cpu()[:, 7].abs().max() == 0
        want = dS[..., :ldt].double().view(B, M - gs, k, ldt // k).sum(-1).permute(0, 2, 1).reshape(B * k, M - gs)
        assert (sums.cpu().double() - want).abs().max() <= 1e-5 * want.abs().max()
        ns = B * k
        outs = []
        for pre in (None, (planes, exps, sums)):
            part, pb = to_device(nan(ns, M, N)), to_device(nan(ns, M))
            HIP.pw_wgrad(B=B, M=M, N=N, T=T, ldt=ldt, G=gdout, G2=gdS, g_split=gs, X=gz, x_mode=PRO_PRELU, x_alpha=gal, partial=part, partial_bias=pb,
                         nsplit=ns, G2_pre=pre)
            device_sync()
            outs.append((part.cpu().double().sum(0), pb.cpu().double().sum(0)))
        u = torch.where(z > 0, z, al * z).double()
        G = torch.cat([dout, dS], 1).double()
        ref = torch.einsum("bmt,bnt->mn", G, u)
        bound = torch.einsum("bmt,bnt->mn", G.abs(), u.abs())
        for (w, bsum), name in zip(outs, ("split in the kernel", "pre-split")):
            assert ((w - ref).abs() <= 2e-6 * bound + 1e-30).all(), name
            assert (bsum - G.sum((0, 2))).abs().max() <= 1e-5 * G.abs().sum((0, 2)).max(), name
    finally:
        sepkernels.set_gemm_arith(prev)


@pytest.mark.parametrize("B,M,N,T,k", [(2, 256, 128, 999, 4), (3, 128, 256, 300, 3), (2, 64, 48, 130, 1), (1, 256, 512, 1000, 8), (3, 32, 20, 500, 2)])
def test_wgrad_sample_aligned_slabs_and_gln_sums_from_them(B, M, N, T, k, arith):
    """The heads' weight gradient as the model's backward takes it: against u = PReLU(z) on nsplit = B * k sample-aligned slabs, then
    sep_gln_bwd_from_wgrad turns slabs + weights into the gLN backward's row sums, the consumer's two totals and the per-sample weight
    gradient of the normalised product; checked against the restatement AND against the sums formed the direct way (dv = W^T g)."""
    ldt = (T + 127) // 128 * 128
    ns = B * k
    z, g = padded(B, N, T, ldt), padded(B, M, T, ldt)
    al = torch.tensor([0.15])
    u = torch.where(z > 0, z, al * z)
    part, pb = _wg_out(ns, M, N)
    kw = dict(B=B, M=M, N=N, T=T, ldt=ldt, G=g, X=z, x_mode=PRO_PRELU, x_alpha=al, partial=part, partial_bias=pb, nsplit=ns)
    _wgrad_both(kw)
    W, gamma, beta, st = rnd(M, N, scale=N ** -0.5), rnd(N) + 1, rnd(N), stats_of(u, T)
    arrive_c, arrive_g = torch.zeros(B, 17, dtype=torch.int32), to_device(torch.zeros(B, 17, dtype=torch.int32))
    args = [part, pb, W, st, gamma, beta, N * float(T), 1e-12, nan(B, M, N), nan(B, N), nan(B, N), zstats(B), None, nan(B, 2), B, M, N, k, 0, 2]
    _sums_both(args, arrive_c, arrive_g)             # `part` / `pb` now hold the emulator's slabs on both sides; first of two products:
    assert not torch.isfinite(args[13]).any()        # ... nothing published yet
    dv = torch.einsum("mn,bmt->bnt", W.double(), g.double())
    R1, R2 = dv.sum(2), (dv * u.double()).sum(2)
    assert (args[9].double() - R1).abs().max() <= 1e-4 * R1.abs().max()
    tot = args[11].sum(1)
    assert torch.allclose(tot[:, 0], (gamma.double() * R1).sum(1), rtol=1e-3, atol=1e-4 * R1.abs().max().item() * N ** 0.5)
    assert torch.allclose(tot[:, 1], (gamma.double() * R2).sum(1), rtol=1e-3, atol=1e-4 * R2.abs().max().item() * N ** 0.5)
    args2 = list(args)
    args2[-2] = 1                                      # accumulate: the second product feeding the same gLN adds its sums and publishes the means
    _sums_both(args2, arrive_c, arrive_g)
    mu = (u[..., :T].double().sum((1, 2)) / (N * T)).view(B, 1)
    rstd = 1.0 / torch.sqrt(((u[..., :T].double() - mu.view(B, 1, 1)) ** 2).sum((1, 2)) / (N * T) + 1e-12).view(B, 1)
    mg = 2 * (gamma.double() * R1).sum(1) / (N * T)
    mgx = 2 * (rstd.view(B) * ((gamma.double() * R2).sum(1) - mu.view(B) * (gamma.double() * R1).sum(1))) / (N * T)
    assert torch.allclose(args2[13][:, 0].double(), mg, rtol=2e-3, atol=1e-6 * mg.abs().max().item() + 1e-12)
    assert torch.allclose(args2[13][:, 1].double(), mgx, rtol=2e-3, atol=2e-3 * mgx.abs().max().item() + 1e-12)


def _sums_both(args, arrive_c, arrive_g):
    """sep_gln_bwd_from_wgrad on both sides with persistent arrival counters (argument 12), everything else compared by `both`"""
    memo_args = list(args)
    memo_args[12] = arrive_c
    gargs = [to_device(v) if torch.is_tensor(v) else v for v in memo_args]
    gargs[12] = arrive_g
    EMU.gln_bwd_from_wgrad(*memo_args)
    HIP.gln_bwd_from_wgrad(*gargs)
    device_sync()
    for i in (8, 9, 10, 13):
        c, g = memo_args[i], gargs[i].cpu()
        assert torch.isfinite(g).all() == torch.isfinite(c).all(), i
        if torch.isfinite(c).all():
            assert (c.double() - g.double()).abs().max() <= 3e-4 * c.double().abs().max() + 1e-30, i
    tc, tg = memo_args[11].sum(1), gargs[11].cpu().sum(1)
    assert (tc - tg).abs().max() <= 3e-4 * tc.abs().max()
    for i in (8, 9, 10, 11, 13):                       # the next call continues from the EMULATOR's state on both sides
        args[i] = memo_args[i]


def test_wgrad_two_sources_gln_prelu(arith):
    B, Bn, Sc, H, T, ns = 2, 128, 64, 256, 640, 5
    ldt = 640
    z = padded(B, H, T, ldt)
    u = torch.where(z > 0, z, 0.1 * z)
    part, pb = _wg_out(ns, Bn + Sc, H)
    _wgrad_both(dict(B=B, M=Bn + Sc, N=H, T=T, ldt=ldt, G=padded(B, Bn, T, ldt), G2=padded(B, Sc, T, ldt), g_split=Bn, X=z,
                     x_mode=PRO_GLN_PRELU, x_stats=stats_of(u, T), x_gamma=rnd(H) + 1, x_beta=rnd(H), x_alpha=torch.tensor([0.1]),
                     count=H * T, eps=1e-12, partial=part, partial_bias=pb, nsplit=ns))


def test_wgrad_latent_product_and_prelu(arith):
    B, n_src, N, LC, T, ns = 2, 3, 64, 16, 500, 4
    ldt = 512
    m, w = padded(B * n_src, N, T, ldt), padded(B, N, T, ldt)
    part, _ = _wg_out(ns, N, LC)
    _wgrad_both(dict(B=B * n_src, M=N, N=LC, T=T, ldt=ldt, G=m, Gaux=w, g_mul=1, g_div=n_src, X=padded(B * n_src, LC, T, ldt),
                     partial=part, nsplit=ns))
    part, pb = _wg_out(ns, 192, 64)
    _wgrad_both(dict(B=B, M=192, N=64, T=T, ldt=ldt, G=padded(B, 192, T, ldt), X=padded(B, 64, T, ldt), x_mode=PRO_PRELU,
                     x_alpha=torch.tensor([0.3]), partial=part, partial_bias=pb, nsplit=ns))
    part, pb = _wg_out(ns, 64, 128)
    Xw = padded(B, 128, T, ldt)
    _wgrad_both(dict(B=B, M=64, N=128, T=T, ldt=ldt, G=padded(B, 64, T, ldt), X=Xw, x_mode=PRO_GLN, x_stats=stats_of(Xw, T),
                     x_gamma=rnd(128) + 1, x_beta=rnd(128), count=128 * T, eps=1e-12, partial=part, partial_bias=pb, nsplit=ns))


def test_reduce_slabs_and_f64():
    src = rnd(6, 1000)
    segs_c = [(src, 0, nan(300), 300, 6, 1000, 0, 1.0), (src, 300, torch.ones(700), 700, 5, 1000, 1, 0.5)]
    segs_g = [(to_device(s), o, to_device(d), n, k, st, a, sc) for (s, o, d, n, k, st, a, sc) in segs_c]
    EMU.reduce_slabs(segs_c)
    HIP.reduce_slabs(segs_g)
    for c, g in zip(segs_c, segs_g):
        assert torch.allclose(c[2], g[2].cpu(), rtol=1e-5, atol=1e-6)
    both("f64_to_f32", [torch.tensor([1.25, -3.5], dtype=torch.float64), torch.tensor([1.0, 1.0]), 2, 1])


# ------------------------------------------------------------------------------------------- depthwise
@pytest.mark.parametrize("T,d", [(3999, 1), (3999, 2), (3999, 128), (300, 8), (1030, 64), (2100, 256)])
def test_dwconv_fwd_bwd(T, d):
    B, C = 2, 8
    ldt = (T + 127) // 128 * 128
    a = padded(B, C, T, ldt)
    a1, a2 = torch.tensor([0.25]), torch.tensor([0.1])
    u1 = torch.where(a > 0, a, a1 * a)
    st1 = stats_of(u1, T)
    g1, b1, wd, bd = rnd(C) + 1, rnd(C), rnd(C, 1, 3), rnd(C)
    z = nan(B, C, ldt)
    st2 = zstats(B)
    both("dwconv_fwd", [a, st1, g1, b1, a1, wd, bd, a2, z, st2, B, C, T, ldt, d, 1e-12])
    # backward on the emulator's z / stats2 (identical inputs for both)
    dv2 = padded(B, C, T, ldt)
    ntile = (ldt + 1023) // 1024
    DV1, RP, BACC, ARR, BSUM = 13, 14, 15, 16, 17
    args = [dv2, z, a, st1, g1, b1, a1, st2, rnd(C) + 1, a2, rnd(B, 2, scale=0.01), wd, bd, nan(B, C, ldt), nan(B, C, ntile, 8), zstats(B),
            torch.zeros(B, 17, dtype=torch.int32), nan(B, 2), B, C, T, ldt, d, 1e-12]
    EMU.dwconv_bwd(*args)
    bc = args[BACC].sum(1)                                                              # gLN1's gamma-weighted totals (slots -> totals)
    rc = args[RP].double().sum(2)                                                       # per-tile partials -> per-row totals
    assert torch.allclose(bc[:, 0], (g1.view(1, C).double() * rc[..., 0]).sum(1), rtol=1e-5, atol=1e-6 * bc.abs().max().item())

    def fresh(bias, mode):
        out = list(args)
        out[12] = bias
        out[DV1], out[RP] = nan(B, C, ldt), nan(B, C, ntile, 8)
        out[BACC], out[ARR], out[BSUM] = {"publish": (zstats(B), torch.zeros(B, 17, dtype=torch.int32), nan(B, 2)), "sums": (zstats(B), None, None),
                                          "none": (None, None, None)}[mode]
        return [to_device(v) if torch.is_tensor(v) else v for v in out]

    # with the bias z is formed again from `a` (what the Conv-TasNet step does), without it z is read
    for bias in (bd, None):
        gargs = fresh(bias, "publish")
        HIP.dwconv_bwd(*gargs)
        device_sync()
        assert torch.isfinite(gargs[DV1]).all() and torch.isfinite(gargs[RP]).all()
        assert (args[DV1] - gargs[DV1].cpu()).abs().max() <= 2e-4 * args[DV1].abs().max()
        rg = gargs[RP].cpu().double().sum(2)
        assert (rc - rg).abs().max() <= 3e-4 * rc.abs().max()
        bg = gargs[BACC].cpu().sum(1)
        assert (bc - bg).abs().max() <= 3e-4 * bc.abs().max()
        # the sample's last workgroup published the two means (and every workgroup arrived exactly once)
        arrived = gargs[ARR].cpu()
        units = arrived[:, :16].sum(1)                     # workgroups (rows, or (channel, tile) units) of every sample: each arrived exactly once
        assert bool((units == units[0]).all()) and int(units[0]) >= C
        assert bool((arrived[:, 16] == min(16, int(units[0]))).all())
        assert (args[BSUM] - gargs[BSUM].cpu()).abs().max() <= 3e-4 * args[BSUM].abs().max()
        # sums only (the consumer forms the means: what the Conv-TasNet step uses)
        g3 = fresh(bias, "sums")
        HIP.dwconv_bwd(*g3)
        device_sync()
        assert (g3[BACC].cpu().sum(1) - bc).abs().max() <= 3e-4 * bc.abs().max()
        assert (args[DV1] - g3[DV1].cpu()).abs().max() <= 2e-4 * args[DV1].abs().max()
        # without the gLN1 outputs (stand-alone use)
        g2_ = fresh(bias, "none")
        HIP.dwconv_bwd(*g2_)
        device_sync()
        assert (args[DV1] - g2_[DV1].cpu()).abs().max() <= 2e-4 * args[DV1].abs().max()


@pytest.mark.parametrize("Kw,stride,pad,dil,Tin", [(3, 1, 1, 1, 300), (5, 2, 4, 2, 257), (4, 4, 0, 1, 64), (3, 1, 8, 8, 1000), (16, 8, 0, 1, 403)])
def test_depthwise_generic(Kw, stride, pad, dil, Tin):
    """sep_depthwise_fwd / bwd_input / bwd_weight (modules/conv.py's depthwise half: any kernel size, stride, padding, dilation)"""
    B, C = 2, 24
    Tout = (Tin + 2 * pad - dil * (Kw - 1) - 1) // stride + 1
    x, w, bias = rnd(B, C, Tin), rnd(C, 1, Kw), rnd(C)
    both("depthwise_fwd", [x, w, bias, nan(B, C, Tout), B, C, Tin, Tout, Kw, stride, pad, dil])
    both("depthwise_fwd", [x, w, None, nan(B, C, Tout), B, C, Tin, Tout, Kw, stride, pad, dil])
    dy = rnd(B, C, Tout)
    both("depthwise_bwd_input", [dy, w, nan(B, C, Tin), B, C, Tin, Tout, Kw, stride, pad, dil])
    both("depthwise_bwd_weight", [dy, x, nan(B, C, Kw + 1), B, C, Tin, Tout, Kw, stride, pad, dil])


@pytest.mark.parametrize("Kw,dil,ldt,causal", [(3, 1, 384, True), (3, 2, 384, True), (3, 4, 512, True), (3, 128, 4096, True), (3, 64, 1024, False), (3, 1, 256, False),
                                                 (5, 2, 384, True), (3, 3, 260, True)])
def test_depthwise_tcn_geometry(Kw, dil, ldt, causal):
    """sep_depthwise_* the way the TCN layers of the staged (causal) path call them: rows of the workspace stride, Tout = Tin, all the
    zero padding in front ((Kw - 1) d, causal) or the smaller half of it.  Kw = 3 takes the float4 row kernels (aligned taps for d % 4 == 0,
    the row through LDS otherwise), Kw = 5 the generic ones."""
    B, C = 2, 24
    pad = (Kw - 1) * dil if causal else (Kw - 1) * dil // 2
    x, w, bias = rnd(B, C, ldt), rnd(C, 1, Kw), rnd(C)
    both("depthwise_fwd", [x, w, bias, nan(B, C, ldt), B, C, ldt, ldt, Kw, 1, pad, dil])
    both("depthwise_fwd", [x, w, None, nan(B, C, ldt), B, C, ldt, ldt, Kw, 1, pad, dil])
    dy = rnd(B, C, ldt)
    both("depthwise_bwd_input", [dy, w, nan(B, C, ldt), B, C, ldt, ldt, Kw, 1, pad, dil])
    both("depthwise_bwd_weight", [dy, x, nan(B, C, Kw + 1), B, C, ldt, ldt, Kw, 1, pad, dil])


@pytest.mark.parametrize("nq,ntile", [(2, 8), (8, 4), (8, 1), (2, 64)])
def test_gln_bwd_finalize(nq, ntile):
    B, C = 3, 96
    rp = rnd(B, C, ntile, nq)
    x = rnd(B, C, 50)
    st = stats_of(x, 50)
    pextra = nan(B * 4 * C + B + B * C) if nq == 8 else None
    both("gln_bwd_finalize", [rp, ntile, nq, st, rnd(C) + 1, C * 50.0, 1e-12, nan(B, 2), nan(B, C), nan(B, C), pextra, B, C])
    both("gln_bwd_finalize", [rp, ntile, nq, st, rnd(C) + 1, C * 50.0, 1e-12, None, nan(B, C), nan(B, C), pextra, B, C])      # without the means


def test_gln_bwd_finalize_batch():
    """several gLNs' second stages in one launch per stage (what the Conv-TasNet step flushes at the end of its backward pass): mixed nq,
    sizes, with and without the means"""
    def seg(B, C, ntile, nq, with_bsum):
        x = rnd(B, C, 50)
        return [rnd(B, C, ntile, nq), ntile, nq, stats_of(x, 50), rnd(C) + 1, C * 50.0, 1e-12, nan(B, 2) if with_bsum else None, nan(B, C), nan(B, C),
                nan(B * 4 * C + B + B * C) if nq == 8 else None, B, C]
    # rows of 32, 16, 8, 128 floats (several rows per wave), 24 (not a power of two of float4s: a wave per row) and 256 (a whole wave of float4s)
    segs = [seg(3, 96, 4, 8, False), seg(2, 40, 8, 2, True), seg(1, 130, 1, 8, True), seg(4, 16, 64, 2, False), seg(2, 33, 3, 8, True), seg(1, 9, 32, 8, False)]
    gsegs = [[to_device(v) if torch.is_tensor(v) else v for v in sg] for sg in segs]
    EMU.gln_bwd_finalize_batch([tuple(sg) for sg in segs])
    HIP.gln_bwd_finalize_batch([tuple(sg) for sg in gsegs])
    device_sync()
    for sg, gg in zip(segs, gsegs):
        for i in (7, 8, 9, 10):
            if sg[i] is None:
                continue
            c, g = sg[i], gg[i].cpu()
            if i == 10:                      # the trailing B*C floats are scratch of the two-stage form
                n = sg[11] * 4 * sg[12] + sg[11]
                c, g = c[:n], g[:n]
            assert torch.isfinite(g).all(), i
            assert (c.double() - g.double()).abs().max() <= 2e-4 * c.double().abs().max() + 1e-30, i


@pytest.mark.parametrize("relu", [0, 1])
def test_head_bwd(relu):
    B, C, T, ldt = 2, 64, 300, 384
    w = padded(B, C, T, ldt)
    both("head_bwd", [padded(B, C, T, ldt), w, padded(B, C, T, ldt), stats_of(w, T), rnd(C) + 1, rnd(B, 2, scale=0.01), B, C, T, ldt, C * float(T), 1e-12, relu])


# ------------------------------------------------------------------------------------------- decoder
@pytest.mark.parametrize("n_src,Cout,L,S,pad_left,latent", [(2, 1, 16, 8, 0, True), (3, 1, 16, 8, 3, False), (5, 1, 16, 8, 0, False), (4, 1, 16, 8, 5, True),
                                                            (2, 2, 20, 10, 4, True), (2, 1, 2, 1, 0, False), (1, 1, 64, 16, 0, False)])
def test_decoder_fwd_bwd(n_src, Cout, L, S, pad_left, latent):
    B, N, F = 2, 64, 300
    ldt = 384
    Tpad = S * (F - 1) + L
    Tout = Tpad - 2 * pad_left - (1 if pad_left else 0)
    w = padded(B, N, F, ldt)
    m = torch.sigmoid(padded(B, n_src * N, F, ldt))
    m[..., F:] = 0
    D = rnd(N, Cout, L)
    both("decoder_fwd", [w, m, D, nan(B, n_src, Cout, Tout), nan(B, n_src, N, ldt) if latent else None, B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left])
    both("decoder_bwd", [rnd(B, n_src, Cout, Tout), w, m, D, nan(B, n_src * N, ldt), nan(B, N, ldt), B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left])
    both("decoder_bwd", [rnd(B, n_src, Cout, Tout), w, m, D, nan(B, n_src * N, ldt), nan(B, N, ldt), B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left],
         dict(raw_mask=1))


@pytest.mark.parametrize("B,C,T", [(2, 128, 300), (1, 1024, 3999), (3, 50, 64), (2, 7, 1)])
def test_softmax_over_channels(B, C, T):
    """mask_nonlinear='softmax': nn.Softmax(dim=1) over the channel rows of every frame, in place; padding frames zeroed
    (they arrive as arbitrary finite values from a GEMM without the pad mask)."""
    ldt = (T + 127) // 128 * 128
    y = rnd(B, C, ldt, scale=3.0)
    both("softmax_ch_fwd", [y, B, C, T, ldt], tol=1e-5)
    probs = torch.zeros(B, C, ldt)
    probs[..., :T] = torch.softmax(rnd(B, C, T, scale=2.0), dim=1)
    both("softmax_ch_bwd", [probs, rnd(B, C, ldt), B, C, T, ldt], tol=1e-5)


# ------------------------------------------------------------------------------------------- stand-alone gLN / repack
def test_gln_standalone_and_repack():
    B, C, T, ldt = 3, 20, 1501, 1504
    x = padded(B, C, T, ldt) + 0.5
    x[..., T:] = 0
    st = zstats(B)
    both("gln_stats", [x, st, B, C, T, ldt])
    gamma, beta = rnd(C) + 1, rnd(C)
    both("gln_apply", [x, st, gamma, beta, nan(B, C, ldt), B, C, T, ldt, C * float(T), 1e-12])
    dy = padded(B, C, T, ldt)
    ntile = (ldt + 1023) // 1024
    rp, rpg = nan(B, C, ntile, 2), to_device(nan(B, C, ntile, 2))
    EMU.gln_bwd_rowsums(dy, x, rp, B, C, T, ldt)
    HIP.gln_bwd_rowsums(to_device(dy), to_device(x), rpg, B, C, T, ldt)
    assert torch.isfinite(rpg).all()          # per-tile partials differ by construction: compare the row totals
    assert (rp.double().sum(2) - rpg.cpu().double().sum(2)).abs().max() <= 2e-4 * rp.abs().max()
    both("gln_bwd_apply", [dy, x, st, gamma, rnd(B, 2, scale=0.01), nan(B, C, ldt), B, C, T, ldt, C * float(T), 1e-12])
    both("repack", [rnd(B * C, T), T, nan(B * C, ldt), ldt, B * C, T])


# ------------------------------------------------------------------------------------------- cumulative layer norm
@pytest.mark.parametrize("B,C,T", [(2, 24, 203), (3, 128, 3999), (1, 512, 5003)])
def test_cln_fwd_bwd(B, C, T):
    """sep_cln_fwd / sep_cln_bwd against the float64 composition of reference src/modules/norm.py:58-101 and its autograd
    backward (T > 1024 and > 4096: the prefix / suffix scans carry across tiles; ldt > T: pad frames come out as zeros)."""
    ldt = (T + 3) // 4 * 4
    torch.manual_seed(B * 1000 + C)
    x = torch.zeros(B, C, ldt)
    x[..., :T] = torch.randn(B, C, T) * torch.linspace(0.3, 2.5, T) + 0.4          # non-stationary, non-zero mean
    gamma, beta, dy = torch.randn(C) + 1, torch.randn(C), torch.zeros(B, C, ldt)
    dy[..., :T] = torch.randn(B, C, T)
    eps = 1e-12
    x64 = x[..., :T].double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    n = torch.arange(1, T + 1, dtype=torch.float64) * C
    m = x64.sum(1).cumsum(1) / n
    v = (x64 * x64).sum(1).cumsum(1) / n - m * m
    y64 = (x64 - m.unsqueeze(1)) / (v.sqrt().unsqueeze(1) + eps) * g64.view(1, C, 1) + b64.view(1, C, 1)
    (y64 * dy[..., :T].double()).sum().backward()
    f32 = dict(device=device_name(), dtype=torch.float32)
    y, mean, rstd = torch.full((B, C, ldt), float("nan"), **f32), torch.empty(B, ldt, **f32), torch.empty(B, ldt, **f32)
    ws = torch.empty((HIP.cln_ws_bytes(B, C, T, ldt) + 7) // 8, device=device_name(), dtype=torch.float64)
    HIP.cln_fwd(to_device(x), to_device(gamma), to_device(beta), y, mean, rstd, ws, B, C, T, ldt, eps)
    dx, pg, pb = torch.full((B, C, ldt), float("nan"), **f32), torch.empty(B, C, **f32), torch.empty(B, C, **f32)
    HIP.cln_bwd(to_device(dy), to_device(x), to_device(gamma), mean, rstd, dx, pg, pb, ws, B, C, T, ldt, eps)
    device_sync()
    y, dx = y.cpu(), dx.cpu()
    assert torch.isfinite(y).all() and torch.isfinite(dx).all()
    assert (y[..., T:] == 0).all() and (dx[..., T:] == 0).all()
    assert (mean.cpu().double()[:, :T] - m.detach()).abs().max() <= 2e-6 * (1 + m.detach().abs().max())
    assert (y[..., :T].double() - y64.detach()).abs().max() <= 2e-5 * y64.detach().abs().max()
    assert (dx[..., :T].double() - x64.grad).abs().max() <= 5e-5 * x64.grad.abs().max()
    assert (pg.cpu().double().sum(0) - g64.grad).abs().max() <= 5e-5 * g64.grad.abs().max()
    assert (pb.cpu().double().sum(0) - b64.grad).abs().max() <= 5e-5 * b64.grad.abs().max()
    # and the emulator's restatement of the same contract
    ye, me, re_ = torch.empty(B, C, ldt), torch.empty(B, ldt), torch.empty(B, ldt)
    EMU.cln_fwd(x, gamma, beta, ye, me, re_, None, B, C, T, ldt, eps)
    assert (ye - y).abs().max() <= 2e-5 * ye.abs().max()
    dxe, pge, pbe = torch.empty(B, C, ldt), torch.empty(B, C), torch.empty(B, C)
    EMU.cln_bwd(dy, x, gamma, me, re_, dxe, pge, pbe, None, B, C, T, ldt, eps)
    assert (dxe - dx).abs().max() <= 5e-5 * dxe.abs().max() and (pge - pg.cpu()).abs().max() <= 5e-5 * pge.abs().max()


@pytest.mark.parametrize("B,C,T,a", [(2, 24, 203, 0.25), (2, 96, 3999, -0.3), (1, 48, 1030, 0.0)])
def test_prelu_cln_fwd_bwd(B, C, T, a):
    """sep_cln_fwd / sep_cln_bwd with the PReLU in front of the norm folded in (alpha != NULL): nonlinear1d -> norm1d of the causal TCN
    layers (reference tdcn.py:113-116, 182-186) against torch's float64 autograd, gradients at x, gamma, beta and the slope."""
    ldt = (T + 127) // 128 * 128
    torch.manual_seed(B * 1000 + C)
    x = torch.zeros(B, C, ldt)
    x[..., :T] = torch.randn(B, C, T) * torch.linspace(0.3, 2.5, T) + 0.1
    gamma, beta, dy = torch.randn(C) + 1, torch.randn(C), torch.zeros(B, C, ldt)
    dy[..., :T] = torch.randn(B, C, T)
    alpha = torch.tensor([a])
    eps = 1e-12
    x64 = x[..., :T].double().requires_grad_(True)
    g64, b64, a64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), alpha.double().requires_grad_(True)
    u64 = torch.where(x64 > 0, x64, a64 * x64)
    n = torch.arange(1, T + 1, dtype=torch.float64) * C
    m = u64.sum(1).cumsum(1) / n
    v = (u64 * u64).sum(1).cumsum(1) / n - m * m
    y64 = (u64 - m.unsqueeze(1)) / (v.sqrt().unsqueeze(1) + eps) * g64.view(1, C, 1) + b64.view(1, C, 1)
    (y64 * dy[..., :T].double()).sum().backward()
    f32 = dict(device=device_name(), dtype=torch.float32)
    y, mean, rstd = torch.full((B, C, ldt), float("nan"), **f32), torch.empty(B, ldt, **f32), torch.empty(B, ldt, **f32)
    ws = torch.empty((HIP.cln_ws_bytes(B, C, T, ldt) + 7) // 8, device=device_name(), dtype=torch.float64)
    HIP.cln_fwd(to_device(x), to_device(gamma), to_device(beta), y, mean, rstd, ws, B, C, T, ldt, eps, alpha=to_device(alpha))
    dx, pg, pb, pa = torch.full((B, C, ldt), float("nan"), **f32), torch.empty(B, C, **f32), torch.empty(B, C, **f32), torch.empty(B, C, **f32)
    HIP.cln_bwd(to_device(dy), to_device(x), to_device(gamma), mean, rstd, dx, pg, pb, ws, B, C, T, ldt, eps, alpha=to_device(alpha), dalpha_part=pa)
    device_sync()
    y, dx = y.cpu(), dx.cpu()
    assert (y[..., T:] == 0).all() and (dx[..., T:] == 0).all()
    assert (y[..., :T].double() - y64.detach()).abs().max() <= 2e-5 * y64.detach().abs().max()
    assert (dx[..., :T].double() - x64.grad).abs().max() <= 5e-5 * x64.grad.abs().max()
    assert (pg.cpu().double().sum(0) - g64.grad).abs().max() <= 5e-5 * g64.grad.abs().max()
    assert (pb.cpu().double().sum(0) - b64.grad).abs().max() <= 5e-5 * b64.grad.abs().max()
    assert abs(pa.cpu().double().sum().item() - a64.grad.item()) <= 2e-4 * (x64.grad.abs() * x64.detach().abs()).sum().item() / max(1, B * C) ** 0.5 + 1e-6
    ye, me, re_ = torch.empty(B, C, ldt), torch.empty(B, ldt), torch.empty(B, ldt)
    EMU.cln_fwd(x, gamma, beta, ye, me, re_, None, B, C, T, ldt, eps, alpha=alpha)
    assert (ye - y).abs().max() <= 2e-5 * ye.abs().max()
    dxe, pge, pbe, pae = torch.empty(B, C, ldt), torch.empty(B, C), torch.empty(B, C), torch.empty(B, C)
    EMU.cln_bwd(dy, x, gamma, me, re_, dxe, pge, pbe, None, B, C, T, ldt, eps, alpha=alpha, dalpha_part=pae)
    assert (dxe - dx).abs().max() <= 5e-5 * dxe.abs().max() and (pae - pa.cpu()).abs().max() <= 2e-4 * pae.abs().max() + 1e-5


@pytest.mark.parametrize("N,L,H,D,p_drop", [(3, 250, 4, 16, 0.0), (2, 257, 2, 32, 0.0), (1, 320, 1, 16, 0.0), (5, 37, 8, 8, 0.0), (2, 100, 4, 16, 0.25), (1, 130, 2, 32, 0.1),
                                            (4, 33, 8, 32, 0.1), (2, 64, 2, 16, 0.0), (2, 128, 2, 8, 0.1), (3, 1, 2, 8, 0.0), (2, 31, 1, 32, 0.0), (2, 65, 2, 32, 0.0),      # tiles for 64 / 128 steps
                                            (3, 31, 8, 32, 0.1), (2, 20, 4, 16, 0.0), (2, 50, 3, 8, 0.0), (5, 32, 12, 8, 0.0)])      # short sequences, many heads, an odd head count
def test_attention_core_fwd_bwd(N, L, H, D, p_drop):
    """sep_attn_fwd / sep_attn_bwd on the packed projection (N, L, 3, H, D): against torch's float64 softmax(q k^T / sqrt(d)) v and its autograd
    gradients (what nn.MultiheadAttention computes between its projections: reference dptnet.py:505-527), sequences that are not multiples of
    32 or 128 (more than 256 steps: the kernels' second size), more than 320 is refused; with dropout against the same composition under the mask the emulator derives from the seed."""
    torch.manual_seed(N * 100 + L)
    qkv = torch.randn(N, L, 3, H, D) * 1.3
    dout = torch.randn(N, L, H, D)
    scale, seed = D ** -0.5, 0x1234ABCD5678
    t = qkv.double().requires_grad_(True)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    pr = torch.softmax(scale * q @ k.transpose(-1, -2), dim=-1)
    keep, kinv = EMU._attn_keep(N, L, H, p_drop, seed)
    if keep is not None:
        assert abs(keep.double().mean().item() - (1 - p_drop)) < 0.02
        pr = torch.where(keep, pr * kinv, torch.zeros_like(pr))
    o64 = (pr @ v).permute(0, 2, 1, 3)
    (o64 * dout.double()).sum().backward()
    f32 = dict(device=device_name(), dtype=torch.float32)
    o, lse = torch.full((N, L, H, D), float("nan"), **f32), torch.full((N, H, L), float("nan"), **f32)
    dq, delta = torch.full((N, L, 3, H, D), float("nan"), **f32), torch.empty(N, H, L, **f32)
    gq = to_device(qkv)
    HIP.attn_fwd(gq, o, lse, N, L, H, D, scale, p_drop, seed)
    HIP.attn_bwd(gq, o, to_device(dout), lse, delta, dq, N, L, H, D, scale, p_drop, seed)
    device_sync()
    o, dq = o.cpu().double(), dq.cpu().double()
    assert torch.isfinite(o).all() and torch.isfinite(dq).all()
    assert (o - o64.detach()).abs().max() <= 2e-5 * o64.detach().abs().max()
    assert (dq - t.grad).abs().max() <= 5e-5 * t.grad.abs().max()
    oe, le, dqe, de = torch.empty(N, L, H, D), torch.empty(N, H, L), torch.empty(N, L, 3, H, D), torch.empty(N, H, L)
    EMU.attn_fwd(qkv, oe, le, N, L, H, D, scale, p_drop, seed)
    EMU.attn_bwd(qkv, oe, dout, le, de, dqe, N, L, H, D, scale, p_drop, seed)
    assert (oe.double() - o).abs().max() <= 2e-5 * o.abs().max() and (le - lse.cpu()).abs().max() <= 2e-5 * (1 + le.abs().max())
    assert (dqe.double() - dq).abs().max() <= 5e-5 * dq.abs().max()
    with pytest.raises(Exception):
        HIP.attn_fwd(gq, o, lse, N, 321, H, D, scale, 0.0, 0)


@pytest.mark.parametrize("nseq,L,C", [(3, 250, 64), (5, 37, 16), (2, 100, 128), (1, 7, 1024), (4, 258, 64), (2, 1500, 64), (3, 4100, 16)])      # the last two: sliced sequences
def test_gln_tokens_fwd_bwd(nseq, L, C):
    """sep_gln_tokens_fwd / bwd: gLN on token-major (nseq, L, C) rows against nn.functional.group_norm on the transposed tensor in float64
    (what the reference computes: dptnet.py:505-560, norm1d(x.permute(1, 2, 0)) with modules/norm.py:11-29)."""
    torch.manual_seed(nseq * 100 + C)
    x = torch.randn(nseq, L, C) * 1.7 + 0.3
    gamma, beta, dy = torch.randn(C) + 1, torch.randn(C), torch.randn(nseq, L, C)
    eps = 1e-12
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y64 = torch.nn.functional.group_norm(x64.permute(0, 2, 1), 1, g64, b64, eps).permute(0, 2, 1)
    (y64 * dy.double()).sum().backward()
    f32 = dict(device=device_name(), dtype=torch.float32)
    y, stats = torch.full((nseq, L, C), float("nan"), **f32), torch.empty(nseq, 2, **f32)
    nws = HIP.gln_tokens_ws_bytes(nseq, L, C)
    ws = torch.empty((nws + 7) // 8, device=device_name(), dtype=torch.float64) if nws else None
    HIP.gln_tokens_fwd(to_device(x), to_device(gamma), to_device(beta), y, stats, nseq, L, C, eps, ws=ws)
    dx, part = torch.full((nseq, L, C), float("nan"), **f32), torch.full((nseq, 2, C), float("nan"), **f32)
    HIP.gln_tokens_bwd(to_device(dy), to_device(x), to_device(gamma), stats, dx, part, nseq, L, C, ws=ws)
    device_sync()
    assert (y.cpu().double() - y64.detach()).abs().max() <= 2e-5 * y64.detach().abs().max()
    assert (dx.cpu().double() - x64.grad).abs().max() <= 5e-5 * x64.grad.abs().max()
    assert (part.cpu().double()[:, 0].sum(0) - g64.grad).abs().max() <= 5e-5 * g64.grad.abs().max()
    assert (part.cpu().double()[:, 1].sum(0) - b64.grad).abs().max() <= 5e-5 * b64.grad.abs().max()
    ye, se = torch.empty(nseq, L, C), torch.empty(nseq, 2)
    EMU.gln_tokens_fwd(x, gamma, beta, ye, se, nseq, L, C, eps)
    dxe, pe = torch.empty(nseq, L, C), torch.empty(nseq, 2, C)
    EMU.gln_tokens_bwd(dy, x, gamma, se, dxe, pe, nseq, L, C)
    assert (ye - y.cpu()).abs().max() <= 2e-5 * ye.abs().max() and (dxe - dx.cpu()).abs().max() <= 5e-5 * dxe.abs().max()
    assert (pe - part.cpu()).abs().max() <= 5e-5 * pe.abs().max()


@pytest.mark.parametrize("rows,C,with_res,p_drop", [(1000, 256, True, 0.0), (33, 256, True, 0.1), (517, 64, False, 0.0), (9, 1024, True, 0.5),
                                                    (130, 300, True, 0.25), (5, 4, True, 0.0), (9000, 128, True, 0.1)])
def test_rownorm_fwd_bwd(rows, C, with_res, p_drop):
    """sep_rownorm_fwd / bwd: LayerNorm_C(x + dropout(res)) on token-major rows against nn.functional.layer_norm in float64 (what the
    reference's nn.TransformerEncoderLayer computes: sepformer.py:395-520, `norm1(x + dropout1(...))`), the dropout mask taken from the
    kernel's own hash (tests/emulator.py::_rownorm_keep); then the emulator call by call."""
    torch.manual_seed(rows + C)
    seed = 0x1234567 * (rows + 1) + (C << 40)
    x = torch.randn(rows, C) * 1.7 + 0.3
    res = torch.randn(rows, C) * 0.8 if with_res else None
    gamma, beta, dy = torch.randn(C) + 1, torch.randn(C), torch.randn(rows, C)
    eps = 1e-5
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    s64 = x64
    if with_res:
        r64 = res.double().requires_grad_(True)
        keep, kinv = EMU._rownorm_keep(rows * C, p_drop, seed)
        # (the kernel scales in fp32: res * fp32(1 / (1 - p)))
        s64 = x64 + (r64 if keep is None else torch.where(keep.view(rows, C), r64 * float(torch.tensor(kinv, dtype=torch.float32)), torch.zeros_like(r64)))
        if keep is not None and rows * C >= 4096:
            assert abs(keep.double().mean().item() - (1 - p_drop)) < 0.03
    y64 = torch.nn.functional.layer_norm(s64, (C,), g64, b64, eps)
    (y64 * dy.double()).sum().backward()
    f32 = dict(device=device_name(), dtype=torch.float32)
    nparts = HIP.rownorm_parts(rows, C)
    assert nparts == EMU.rownorm_parts(rows, C)
    y, stat = torch.full((rows, C), float("nan"), **f32), torch.full((rows, 2), float("nan"), **f32)
    s = torch.full((rows, C), float("nan"), **f32) if with_res else None
    HIP.rownorm_fwd(to_device(x), to_device(res) if with_res else None, to_device(gamma), to_device(beta), s, y, stat, rows, C, eps, p_drop, seed)
    ds, part = torch.full((rows, C), float("nan"), **f32), torch.full((nparts, 2, C), float("nan"), **f32)
    dres = torch.full((rows, C), float("nan"), **f32) if p_drop > 0 else None
    HIP.rownorm_bwd(to_device(dy), s if with_res else to_device(x), to_device(gamma), stat, ds, dres, part, rows, C, p_drop, seed)
    device_sync()
    assert (y.cpu().double() - y64.detach()).abs().max() <= 2e-5 * y64.detach().abs().max()
    assert (ds.cpu().double() - x64.grad).abs().max() <= 5e-5 * x64.grad.abs().max()
    if with_res:
        assert (s.cpu().double() - s64.detach()).abs().max() <= 1e-6 * s64.detach().abs().max()
        got = dres if p_drop > 0 else ds
        assert (got.cpu().double() - r64.grad).abs().max() <= 5e-5 * r64.grad.abs().max()
    assert (part.cpu().double()[:, 0].sum(0) - g64.grad).abs().max() <= 5e-5 * max(g64.grad.abs().max().item(), rows ** 0.5)
    assert (part.cpu().double()[:, 1].sum(0) - b64.grad).abs().max() <= 5e-5 * max(b64.grad.abs().max().item(), rows ** 0.5)
    ye, se, sse = torch.empty(rows, C), torch.empty(rows, 2), (torch.empty(rows, C) if with_res else None)
    EMU.rownorm_fwd(x, res, gamma, beta, sse, ye, se, rows, C, eps, p_drop, seed)
    dse, pe, dre = torch.empty(rows, C), torch.empty(nparts, 2, C), (torch.empty(rows, C) if p_drop > 0 else None)
    EMU.rownorm_bwd(dy, sse if with_res else x, gamma, se, dse, dre, pe, rows, C, p_drop, seed)
    assert (ye - y.cpu()).abs().max() <= 2e-5 * ye.abs().max() and (dse - ds.cpu()).abs().max() <= 5e-5 * dse.abs().max()
    assert (se - stat.cpu()).abs().max() <= 2e-5 * se.abs().max()
    assert (pe - part.cpu()).abs().max() <= 5e-5 * max(pe.abs().max().item(), 1.0)
    if p_drop > 0:
        # zero exactly where the mask drops (an element of ds itself may round to zero on one side only: the mask is compared, not the zeros)
        dropped = ~EMU._rownorm_keep(rows * C, p_drop, seed)[0].view(rows, C)
        assert (dres.cpu()[dropped] == 0).all() and (dre[dropped] == 0).all()
        assert (dre - dres.cpu()).abs().max() <= 5e-5 * dre.abs().max()
    if with_res:                                               # inference: the sum is not kept
        y2, stat2 = torch.full((rows, C), float("nan"), **f32), torch.full((rows, 2), float("nan"), **f32)
        HIP.rownorm_fwd(to_device(x), to_device(res), to_device(gamma), to_device(beta), None, y2, stat2, rows, C, eps, p_drop, seed)
        device_sync()
        assert torch.equal(y2.cpu(), y.cpu()) and torch.equal(stat2.cpu(), stat.cpu())
    with pytest.raises(Exception):
        HIP.rownorm_fwd(to_device(x), None, to_device(gamma), to_device(beta), None, y, stat, rows, C, eps, 0.5, seed)      # dropout without a branch
    with pytest.raises(Exception):
        HIP.rownorm_fwd(to_device(x), None, to_device(gamma), to_device(beta), ds, y, stat, rows, C, eps, 0.0, seed)        # a place for the sum without a branch


@pytest.mark.parametrize("n,p_drop", [(4096, 0.0), (1028, 0.1), (4, 0.5), (3000000, 0.1)])
def test_relu_drop_fwd_bwd(n, p_drop):
    """sep_relu_drop_fwd / bwd: dropout(relu(h)) of the feed-forward sub-block against the emulator's restatement (same hash), the rate of the
    mask, and the backward pass reading the mask off the forward's output"""
    torch.manual_seed(n)
    seed = 0xABCDEF0123 + n
    h, dy = torch.randn(n), torch.randn(n)
    f32 = dict(device=device_name(), dtype=torch.float32)
    a, dh = torch.full((n,), float("nan"), **f32), torch.full((n,), float("nan"), **f32)
    HIP.relu_drop_fwd(to_device(h), a, n, p_drop, seed)
    HIP.relu_drop_bwd(to_device(dy), a, dh, n, p_drop)
    device_sync()
    ae, dhe = torch.empty(n), torch.empty(n)
    EMU.relu_drop_fwd(h, ae, n, p_drop, seed)
    EMU.relu_drop_bwd(dy, ae, dhe, n, p_drop)
    assert torch.equal(ae == 0, a.cpu() == 0) and (ae - a.cpu()).abs().max() <= 1e-6 * ae.abs().max()
    assert torch.equal(dhe == 0, dh.cpu() == 0) and (dhe - dh.cpu()).abs().max() <= 1e-6 * dhe.abs().max()
    kinv = 1.0 / (1.0 - p_drop)
    kept = a.cpu() != 0
    assert torch.all(h[kept] > 0) and (a.cpu()[kept] - h[kept] * kinv).abs().max() <= 1e-6 * h.abs().max() * kinv
    if n >= 4096:
        assert abs(kept.double().sum().item() / (h > 0).double().sum().item() - (1 - p_drop)) < 0.03
    with pytest.raises(Exception):
        HIP.relu_drop_fwd(to_device(h), a, n + 1, p_drop, seed)


# ------------------------------------------------------------------------------------------- losses / optimiser
@pytest.mark.parametrize("n,all_pairs", [(1, 0), (2, 1), (4, 1), (3, 0)])
def test_sisdr_kernels(n, all_pairs):
    B, T = 3, 32000
    tgt = rnd(B, n, T)
    est = 0.8 * tgt[:, torch.randperm(n, generator=G)] + 0.3 * rnd(B, n, T)
    dots, tt, xx = (torch.zeros(B, n, n, dtype=torch.float64), torch.zeros(B, n, dtype=torch.float64), torch.zeros(B, n, dtype=torch.float64))
    both("sisdr_dots", [est, tgt, dots, tt, xx, B, n, T, all_pairs], tol=1e-6)
    both("sisdr_from_dots", [dots, tt, xx, nan(B, n, n), B, n, all_pairs, 1e-12], tol=1e-5)
    both("sisdr_bwd", [est, tgt, dots, tt, xx, rnd(B, n, n), nan(B, n, T), B, n, T, all_pairs, 1e-12], tol=1e-5)


@pytest.mark.parametrize("n,maximize,use_mean", [(2, 0, 1), (3, 1, 1), (4, 0, 0)])
def test_pit_search(n, maximize, use_mean):
    B = 9
    perms = torch.tensor(list(itertools.permutations(range(n))), dtype=torch.int32)
    both("pit_search", [rnd(B, n, n), perms, perms.size(0), n, B, maximize, use_mean, nan(B), torch.zeros(B, dtype=torch.int64)])


@pytest.mark.parametrize("n,iters,beta", [(3, 10, 1.0), (5, 200, 1.0), (4, 20, 2.0), (10, 5, 0.5)])
def test_sinkhorn(n, iters, beta):
    B = 4
    C = rnd(B, n, n, scale=3.0)
    zw = torch.full((B, 2 * iters + 1, n, n), float("nan"), dtype=torch.float64)
    both("sinkhorn_fwd", [C, zw, nan(B), nan(B, n, n), B, n, beta, iters], tol=1e-5)
    both("sinkhorn_bwd", [C, zw, rnd(B), nan(B, n, n), B, n, beta, iters], tol=1e-5)


@pytest.mark.parametrize("rows,T", [(8, 32000), (6, 31999), (1, 5), (300, 257), (16, 88200)])
def test_rowdiff_sums_and_bwd(rows, T):
    x, t = rnd(rows, T), rnd(rows, T, scale=0.7)
    x[0, :min(T, 3)] = t[0, :min(T, 3)]                        # exact ties: sign(0) = 0
    both("rowdiff_sums", [x, t, nan(rows, 3).double(), rows, T], tol=1e-5)
    ca, cs = rnd(rows), rnd(rows)
    both("rowdiff_bwd", [x, t, ca, cs, nan(rows, T), rows, T], tol=1e-6)
    both("rowdiff_bwd", [x, t, ca, None, nan(rows, T), rows, T], tol=1e-6)
    both("rowdiff_bwd", [x, t, None, cs, nan(rows, T), rows, T], tol=1e-6)
    # rows that start off a 16-byte boundary take the scalar path
    xo, to = rnd(rows * T + 1)[1:].reshape(rows, T), rnd(rows * T + 1)[1:].reshape(rows, T)
    gx, gt = torch.empty(rows * T + 1, device=device_name())[1:].reshape(rows, T).copy_(xo), torch.empty(rows * T + 1, device=device_name())[1:].reshape(rows, T).copy_(to)
    sc, sg = torch.empty(rows, 3, dtype=torch.float64), torch.empty(rows, 3, dtype=torch.float64, device=device_name())
    EMU.rowdiff_sums(xo, to, sc, rows, T)
    HIP.rowdiff_sums(gx, gt, sg, rows, T)
    assert (sg.cpu() - sc).abs().max() <= 1e-5 * sc.abs().max()


def test_sqnorm_and_adam():
    n = 100003
    p, g, m, v = rnd(n), rnd(n, scale=3.0), rnd(n, scale=0.1), rnd(n, scale=0.1).abs()
    sq = torch.zeros(1, dtype=torch.float64)
    both("sqnorm", [g, sq, n], tol=1e-6)
    both("adam_step", [p, g, m, v, sq, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 5.0, 0.5, 3], tol=1e-5)
    both("adam_step", [p, g, m, v, sq, n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.0, 1.0, 4], tol=1e-5)


@pytest.mark.parametrize("T,chunk,hop", [(812, 20, 10), (3999, 250, 125), (100, 16, 4), (64, 64, 64)])
def test_segment_overlap_add(T, chunk, hop):
    B, C = 2, 5
    ldt = (T + 127) // 128 * 128
    padding = (hop - (T - chunk) % hop) % hop
    pad_left = padding // 2
    S = (T + padding - chunk) // hop + 1
    x = padded(B, C, T, ldt)
    both("segment", [x, nan(B, C, S, chunk), B * C, T, ldt, S, chunk, hop, pad_left], tol=0.0)
    both("overlap_add", [rnd(B, C, S, chunk), nan(B, C, ldt), B * C, T, ldt, S, chunk, hop, pad_left], tol=1e-6)


# ------------------------------------------------------------------------------------------- LSTM recurrence (DPRNN)
LSTM_KERNELS = {"sixteen": 0x100, "four": 0x200}      # SEP_LSTM_FORCE16 / SEP_LSTM_FORCE4: every case runs on both sweep kernels


# ------------------------------------------------------------------------------------------- chunked features <-> token-major rows
@pytest.mark.parametrize("B,F,S,K", [(2, 64, 5, 250), (1, 48, 3, 33), (3, 7, 2, 1), (2, 64, 257, 50)])
def test_chunk_tokens_layout_pair(B, F, S, K):
    """sep_chunk_to_tokens / sep_tokens_to_chunk against permute + reshape, both sequence axes, ragged 32 x 32 tiles, and as inverses."""
    x = rnd(B, F, S, K)
    for inter in (0, 1):
        y = nan(B * K, S, F) if inter else nan(B * S, K, F)
        both("chunk_to_tokens", [x, y, B, F, S, K, inter], tol=0)
        want = x.permute(0, 3, 2, 1).reshape(B * K, S, F) if inter else x.permute(0, 2, 3, 1).reshape(B * S, K, F)
        assert torch.equal(y, want)
        back = nan(B, F, S, K)
        both("tokens_to_chunk", [y, back, B, F, S, K, inter], tol=0)
        assert torch.equal(back, x)


# ------------------------------------------------------------------------------------------- token-major dense layers
@pytest.mark.parametrize("ntok,K,N", [(1000, 64, 512), (777, 256, 64), (130, 128, 128), (64, 64, 64), (3001, 128, 512)])
def test_linear_forward_and_input_gradient(ntok, K, N):
    """sep_linear_fwd / sep_linear_bwd_input against x @ w.t() + b and dy @ w (ragged last 128-token tile; both column tile widths)."""
    x, w, b1, b2 = rnd(ntok, K), rnd(N, K, scale=K ** -0.5), rnd(N), rnd(N)
    both("linear_fwd", [x, w, b1, b2, nan(ntok, N), ntok, K, N])
    both("linear_fwd", [x, w, None, None, nan(ntok, N), ntok, K, N])
    dy = rnd(ntok, N)
    both("linear_bwd_input", [dy, w, nan(ntok, K), ntok, K, N, 0], tol=2e-4 * (N / 64) ** 0.5)
    both("linear_bwd_input", [dy, w, rnd(ntok, K), ntok, K, N, 1], tol=2e-4 * (N / 64) ** 0.5)


@pytest.mark.parametrize("nseq,L,K,N,shift,nslab", [(32, 20, 64, 512, 0, 7), (9, 31, 128, 512, -1, 3), (9, 31, 128, 512, 1, 5), (5, 250, 256, 64, 0, 40),
                                                    (3, 7, 64, 64, -1, 1), (40, 50, 128, 128, 1, 64)])
def test_linear_weight_gradient(nseq, L, K, N, shift, nslab):
    """sep_linear_bwd_weight: dy.t() @ x in nslab partial sums (slabs beyond the last token are zero), the bias sums, and the x operand
    read one step earlier / later inside sequences of L steps (the h_{t-1} of an LSTM's recurrent weight gradient)."""
    ntok = nseq * L
    dy, x = rnd(ntok, N), rnd(ntok, K)
    part_c, pb_c = nan(nslab, N, K), nan(nslab, N)
    part_g, pb_g = to_device(part_c), to_device(pb_c)
    EMU.linear_bwd_weight(dy, x, K, part_c, pb_c, ntok, K, N, L, shift, nslab)
    HIP.linear_bwd_weight(to_device(dy), to_device(x), K, part_g, pb_g, ntok, K, N, L, shift, nslab)
    device_sync()
    ref = part_c.sum(0)
    assert torch.isfinite(part_g).all() and torch.isfinite(pb_g).all()
    assert (part_g.cpu() - part_c).abs().max() <= 2e-4 * ref.abs().max()                 # slab by slab: the partition is part of the contract
    assert (pb_g.cpu() - pb_c).abs().max() <= 2e-4 * max(1.0, pb_c.sum(0).abs().max().item())
    # ... and against the plain formula
    xs = x.reshape(nseq, L, K)
    if shift:
        z = torch.zeros_like(xs)
        if shift < 0:
            z[:, 1:] = xs[:, :-1]
        else:
            z[:, :-1] = xs[:, 1:]
        xs = z
    want = dy.double().t() @ xs.reshape(ntok, K).double()
    assert (part_g.cpu().double().sum(0) - want).abs().max() <= 2e-4 * want.abs().max()
    HIP.linear_bwd_weight(to_device(dy), to_device(x), K, part_g, None, ntok, K, N, L, shift, nslab)      # without the bias sums
    device_sync()
    # x as the right half of a wider row-major matrix (rows 2K + 4 floats apart): one direction of an interleaved bi-LSTM output
    wide = rnd(ntok, 2 * K + 4)
    wide[:, K + 4:] = x
    wide_g = to_device(wide)
    part2 = to_device(nan(nslab, N, K))
    HIP.linear_bwd_weight(to_device(dy), wide_g[:, K + 4:], 2 * K + 4, part2, None, ntok, K, N, L, shift, nslab)
    device_sync()
    assert (part2.cpu() - part_c).abs().max() <= 2e-4 * ref.abs().max()


@pytest.mark.parametrize("kernel", sorted(LSTM_KERNELS))
@pytest.mark.parametrize("H,nseq,L,reverse", [(16, 5, 7, 0), (32, 37, 23, 1), (64, 16, 40, 0), (128, 50, 31, 1), (128, 33, 250, 0)])
def test_lstm_sweeps(H, nseq, L, reverse, kernel):
    """sep_lstm_fwd / sep_lstm_bwd against the step-by-step CPU restatement (ragged last workgroup: nseq % 16 != 0, nseq % 4 != 0)."""
    reverse |= LSTM_KERNELS[kernel]
    xg = rnd(nseq, L, 4 * H)
    w_hh = rnd(4 * H, H, scale=H ** -0.5)
    h, gates, cst = nan(nseq, L, H), nan(nseq, L, 4 * H), nan(nseq, L, H)
    tol = 2e-4 if L < 100 else 2e-3
    both("lstm_fwd", [xg, w_hh, h, gates, cst, nseq, L, H, reverse], tol=tol)
    # backward on the emulator's saved gates / cell states (identical inputs for both sides)
    both("lstm_bwd", [rnd(nseq, L, H), gates, cst, w_hh, nan(nseq, L, 4 * H), nseq, L, H, reverse], tol=tol)


@pytest.mark.parametrize("kernel", sorted(LSTM_KERNELS))
def test_lstm_sweeps_interleaved_output(kernel):
    """reverse = 2 | SEP_LSTM_INTERLEAVED: h_out and dh_out are one (nseq, L, 2H) buffer, the layout nn.LSTM(bidirectional=True) returns."""
    H, nseq, L = 64, 9, 17
    mode = 2 | 0x400 | LSTM_KERNELS[kernel]
    xg = rnd(2, nseq, L, 4 * H)
    w_hh = rnd(2, 4 * H, H, scale=H ** -0.5)
    h, gates, cst = nan(nseq, L, 2 * H), nan(2, nseq, L, 4 * H), nan(2, nseq, L, H)
    both("lstm_fwd", [xg, w_hh, h, gates, cst, nseq, L, H, mode])
    slabs = nan(2, nseq, L, H)
    EMU.lstm_fwd(xg, w_hh, slabs, None, None, nseq, L, H, 2)
    assert torch.equal(h, torch.cat([slabs[0], slabs[1]], dim=2))
    dh = rnd(nseq, L, 2 * H)
    both("lstm_bwd", [dh, gates, cst, w_hh, nan(2, nseq, L, 4 * H), nseq, L, H, mode])


@pytest.mark.parametrize("kernel", sorted(LSTM_KERNELS))
def test_lstm_sweeps_both_directions_one_launch(kernel, nseq=21, L=40):
    """reverse = 2: two slabs per buffer, grid.y = direction.  (nseq, L: the host simulation of the CPU tier runs a shorter case)"""
    H = 128
    two = 2 | LSTM_KERNELS[kernel]
    xg = rnd(2, nseq, L, 4 * H)
    w_hh = rnd(2, 4 * H, H, scale=H ** -0.5)
    h, gates, cst = nan(2, nseq, L, H), nan(2, nseq, L, 4 * H), nan(2, nseq, L, H)
    both("lstm_fwd", [xg, w_hh, h, gates, cst, nseq, L, H, two])
    both("lstm_bwd", [rnd(2, nseq, L, H), gates, cst, w_hh, nan(2, nseq, L, 4 * H), nseq, L, H, two])



@pytest.mark.parametrize("B,Bn,H,Sc,T,d", [(2, 128, 512, 128, 3999, 4), (2, 128, 512, 128, 3999, 128), (1, 128, 256, 128, 1030, 1)])
def test_tcn_layer_kernel_by_kernel_against_the_oracle(B, Bn, H, Sc, T, d):
    """round-4 verdict item 7: the six big kernels of a TCN layer at the MODEL'S shapes -- conv1, depthwise forward, heads; heads^T, heads
    weight gradient (+ the gLN sums formed from it), depthwise backward, conv1^T (with its d(pre-activation) store-back), conv1 weight
    gradient -- launched one by one exactly as sepkernels/net.py launches them (packed weights, f16x3) and EVERY intermediate compared with
    the oracle's own primitives (oracle.convtasnet_oracle.pointwise_conv / prelu / gln / depthwise_conv) and autograd through them in
    fp64, on the same buffers.  No emulator in the chain: a bug shared by tests/emulator.py and a kernel shows up here."""
    from oracle import convtasnet_oracle as O
    from sepkernels import net as NET
    from sepkernels import ARRIVE_INTS
    prev = sepkernels.set_gemm_arith("f16x3")
    try:
        K = HIP
        eps = 1e-12
        ldt = (T + 127) // 128 * 128
        F = T
        cnt = H * F
        x = padded(B, Bn, T, ldt)
        W1, b1 = rnd(H, Bn, 1, scale=Bn ** -0.5), rnd(H, scale=0.1)
        al1, al2 = torch.tensor([0.25]), torch.tensor([0.1])
        g1, be1, g2, be2 = rnd(H, scale=0.2) + 1, rnd(H, scale=0.2), rnd(H, scale=0.2) + 1, rnd(H, scale=0.2)
        wd, bd = rnd(H, 1, 3, scale=0.5), rnd(H, scale=0.1)
        Wcat, bcat = rnd(Bn + Sc, H, scale=H ** -0.5), rnd(Bn + Sc, scale=0.1)
        dout, dS = padded(B, Bn, T, ldt), padded(B, Sc, T, ldt)

        def close(got, ref, tol, what):
            got = got.detach().cpu().double()
            ref = ref.detach().double()
            assert torch.isfinite(got).all(), what
            diff = (got - ref).abs()
            err = diff.max().item()
            where = tuple(int(i) for i in torch.unravel_index(diff.argmax(), diff.shape)) if diff.dim() else ()
            nbad = int((diff > tol * ref.abs().max()).sum())
            assert err <= tol * (ref.abs().max().item() + 1e-30), "{}: max err {:.3e} vs scale {:.3e} at {} ({} of {} elements above the bound)".format(
                what, err, ref.abs().max().item(), where, nbad, diff.numel())

        dev = lambda t: to_device(t)
        f32 = dict(device=device_name(), dtype=torch.float32)
        X, gW1, gb1, gal1, gal2, gg1, gbe1, gg2, gbe2, gwd, gbd, gWc, gbc, gdout, gdS = (dev(t) for t in (x, W1, b1, al1, al2, g1, be1, g2, be2, wd, bd, Wcat, bcat, dout, dS))
        gWo, gWs = gWc[:Bn], gWc[Bn:]
        pk = K.pack_weights([(gW1.view(H, Bn), H, Bn, 0), (gW1.view(H, Bn), H, Bn, 1), (gWc, Bn + Sc, H, 0), (gWc, Bn + Sc, H, 1)])
        st = dev(torch.zeros(3, B, SLOTS, 2, dtype=torch.float64))
        # ---- forward: conv1 -> depthwise -> heads ------------------------------------------------------------------------------------
        a = dev(nan(B, H, ldt))
        K.pw_gemm(B=B, M=H, K=Bn, T=F, ldt=ldt, A=gW1, A_pk=pk[0], X=X, Y=a, bias=gb1, epi_flags=EPI_STATS_PRELU, epi_alpha=gal1, epi_stats=st[1], eps=eps)
        device_sync()
        assert a[..., T:].abs().max().item() == 0.0
        s1 = st[1].cpu().sum(1)
        z = dev(nan(B, H, ldt))
        K.dwconv_fwd(a, st[1], gg1, gbe1, gal1, gwd, gbd, gal2, z, st[2], B, H, F, ldt, d, eps)
        device_sync()
        xo, skip = dev(nan(B, Bn, ldt)), dev(nan(B, Sc, ldt))
        K.pw_gemm(B=B, M=Bn + Sc, K=H, T=F, ldt=ldt, A=gWc, A_pk=pk[2], X=z, Y=xo, Y2=skip, m_split=Bn, bias=gbc, accumulate=0, epi_flags=EPI_RESIDUAL,
                  epi_res=X, pro_mode=PRO_GLN_PRELU, pro_stats=st[2], pro_gamma=gg2, pro_beta=gbe2, pro_alpha=gal2, count=cnt, eps=eps)
        device_sync()
        # ---- the oracle, fp64, with every intermediate kept.  PReLU is not differentiable at 0: where a pre-activation of the device (fp32
        # products) and of the oracle (fp64) fall on different sides of it -- a handful of the 4 M elements, |value| ~ 1e-6 -- the two slopes
        # differ by 1 - alpha and the gradients behind that element (three taps, or a whole frame of dx) are O(1) apart with nothing wrong on
        # either side.  The oracle therefore takes its PReLU BRANCH from the device's pre-activations (checked below to differ from its own
        # only where the value is within rounding of zero); every value and every other operation is its own.
        dd = lambda t: t.double().clone().requires_grad_(True)
        ox = dd(x[..., :T])
        oW1, ob1, oal1, oal2, og1, obe1, og2, obe2, owd, obd, oWc, obc = (dd(t) for t in (W1, b1, al1, al2, g1, be1, g2, be2, wd, bd, Wcat, bcat))
        oa = O.pointwise_conv(ox, oW1, ob1)
        a_dev, z_dev = a[..., :T].cpu().double(), None
        flip = (a_dev > 0) != (oa.detach() > 0)
        assert not flip.any() or oa.detach()[flip].abs().max() <= 1e-4 * oa.detach().abs().max()
        ou1 = torch.where(a_dev > 0, oa, oal1.reshape(()) * oa)
        assert (ou1.detach() - O.prelu(oa.detach(), oal1.detach())).abs().max() <= 1e-4 * oa.detach().abs().max()
        ov1 = O.gln(ou1, og1, obe1, eps)
        oz = O.depthwise_conv(ov1, owd, obd, d)
        z_dev = z[..., :T].cpu().double()
        flip = (z_dev > 0) != (oz.detach() > 0)
        assert not flip.any() or oz.detach()[flip].abs().max() <= 1e-4 * oz.detach().abs().max()
        ou2 = torch.where(z_dev > 0, oz, oal2.reshape(()) * oz)
        ov2 = O.gln(ou2, og2, obe2, eps)
        oo = O.pointwise_conv(ov2, oWc[:Bn].unsqueeze(-1), obc[:Bn]) + ox
        osk = O.pointwise_conv(ov2, oWc[Bn:].unsqueeze(-1), obc[Bn:])
        for t in (oa, ov1, oz, ov2):
            t.retain_grad()
        ((oo * dout[..., :T].double()).sum() + (osk * dS[..., :T].double()).sum()).backward()
        close(a[..., :T], oa, 2e-4, "conv1")
        sq = (ou1.detach() ** 2).sum((1, 2))
        assert (s1[:, 1] - sq).abs().max() <= 1e-4 * sq.abs().max()                     # the statistics the epilogue adds up: sum of PReLU(a)^2
        assert (s1[:, 0] - ou1.detach().sum((1, 2))).abs().max() <= 1e-4 * sq.sqrt().max() * (H * T) ** 0.5
        close(z[..., :T], oz, 2e-4, "depthwise forward")
        close(xo[..., :T], oo, 2e-4, "heads: output + residual")
        close(skip[..., :T], osk, 2e-4, "heads: skip")
        # ---- backward: heads weight gradient + gLN2 sums, heads^T, depthwise backward, conv1^T, conv1 weight gradient ------------------
        bacc = dev(torch.zeros(3, B, SLOTS, 2, dtype=torch.float64))
        arrive = dev(torch.zeros(3, B, ARRIVE_INTS, dtype=torch.int32))
        bsum = dev(nan(3, B, 2))
        pbeta2, pgamma2 = dev(nan(B, H)), dev(nan(B, H))
        rows = Bn + Sc
        part, pb, ns = NET._wgrad(K, B, F, ldt, eps, f32, rows, H, gdout, z, True, x_mode=PRO_PRELU, x_alpha=gal2, weps=eps, aligned=True, G2=gdS, g_split=Bn)
        dWb = dev(nan(B, rows, H))
        K.gln_bwd_from_wgrad(part, pb, gWc, st[2], gg2, gbe2, cnt, eps, dWb, pbeta2, pgamma2, bacc[2], arrive[2], bsum[2], B, rows, H, ns // B, accumulate=0, products=1)
        device_sync()
        close(dWb.sum(0), oWc.grad, 5e-4, "heads weight gradient")
        close(pb.sum(0), obc.grad, 5e-4, "heads bias gradient")
        close(pgamma2.sum(0), og2.grad, 5e-4, "gLN2 gain gradient")
        close(pbeta2.sum(0), obe2.grad, 5e-4, "gLN2 shift gradient")
        dv2 = dev(nan(B, H, ldt))
        K.pw_gemm(B=B, M=H, K=rows, T=F, ldt=ldt, trans_a=1, A=gWo, A2=gWs, A_pk=pk[3], X=gdout, X2=gdS, k_split=Bn, Y=dv2, eps=eps)
        device_sync()
        close(dv2[..., :T], ov2.grad, 5e-4, "heads^T")
        dv1 = dev(nan(B, H, ldt))
        nt1024 = (ldt + 1023) // 1024
        rp1 = dev(nan(B, H, nt1024, 8))
        K.dwconv_bwd(dv2, z, a, st[1], gg1, gbe1, gal1, st[2], gg2, gal2, bsum[2], gwd, gbd, dv1, rp1, bacc[1], None, None, B, H, F, ldt, d, eps)
        device_sync()
        close(dv1[..., :T], ov1.grad, 5e-4, "depthwise backward")
        dx = dev(nan(B, Bn, ldt))
        dal = dev(torch.zeros(1, dtype=torch.float64))
        K.pw_gemm(B=B, M=Bn, K=H, T=F, ldt=ldt, trans_a=1, A=gW1, A_pk=pk[1], X=dv1, Y=dx, pro_mode=PRO_GLN_BWD, pro_stats=st[1], pro_gamma=gg1, pro_alpha=gal1,
                  pro_aux=a, pro_bacc=bacc[1], pro_store=dv1, pro_dalpha=dal, count=cnt, eps=eps, epi_flags=EPI_RESIDUAL, epi_res=gdout)
        device_sync()
        close(dx[..., :T], ox.grad, 5e-4, "conv1^T + residual")
        close(dv1[..., :T], oa.grad, 5e-4, "conv1^T store-back: d(pre-activation)")
        close(dal, oal1.grad, 2e-3, "PReLU1 slope gradient")
        part, pb, ns = NET._wgrad(K, B, F, ldt, eps, f32, H, Bn, dv1, X, True)
        device_sync()
        close(part.sum(0), oW1.grad[..., 0], 5e-4, "conv1 weight gradient")
        close(pb.sum(0), ob1.grad, 5e-4, "conv1 bias gradient")
        pbeta1, pgamma1, pextra = dev(nan(B, H)), dev(nan(B, H)), dev(nan(B * 4 * H + B + B * H))
        K.gln_bwd_finalize(rp1, nt1024, 8, st[1], gg1, cnt, eps, None, pbeta1, pgamma1, pextra, B, H)
        device_sync()
        close(pgamma1.sum(0), og1.grad, 5e-4, "gLN1 gain gradient")
        close(pbeta1.sum(0), obe1.grad, 5e-4, "gLN1 shift gradient")
        ex = pextra[:B * 4 * H].view(B, 4 * H)
        close(ex[:, :H].sum(0), obd.grad, 5e-4, "depthwise bias gradient")
        close(ex[:, H:].sum(0).view(H, 1, 3), owd.grad, 5e-4, "depthwise weight gradient")
        close(pextra[B * 4 * H:B * 4 * H + B].sum(), oal2.grad.reshape(()), 2e-3, "PReLU2 slope gradient")
    finally:
        sepkernels.set_gemm_arith(prev)


@pytest.mark.parametrize("B,N,Bn,Sc,n_src,T_in,relu", [(2, 512, 128, 128, 2, 2405, False), (1, 512, 128, 128, 2, 2392, True), (2, 256, 128, 128, 3, 1203, False)])
def test_head_and_tail_kernel_by_kernel_against_the_oracle(B, N, Bn, Sc, n_src, T_in, relu):
    """round-5 verdict item 7: the kernels AROUND the TCN at the model's shapes, launched one by one exactly as sepkernels/net.py launches
    them (head_forward / tail_forward / tail_backward / head_backward: packed weights, f16x3) and EVERY intermediate compared with the
    oracle's own primitives (oracle.convtasnet_oracle.encoder / gln / pointwise_conv / prelu / decoder) and autograd through them in fp64:
        encoder (+ReLU, statistics)  ->  gLN-prologue bottleneck product        |  PReLU-prologue mask product + sigmoid  ->  mask * w, decoder
        overlap-add, crop;   backward: unfold + decoder-basis weight gradient, decoder^T (d mask pre-activation, d w through the mask), mask^T
        with the PReLU backward in its epilogue (+ slope gradient), mask weight gradient; bottleneck weight gradient (gLN on its X operand),
        bottleneck^T with row sums, gLN finalisation, head backward (gLN^T + the mask path's d w + ReLU mask), unfold + encoder-basis gradient.
    The TCN between the two halves is replaced by given tensors (a random skip sum, a random gradient at the bottleneck output).  No
    emulator in the chain.  Reference: src/models/conv_tasnet.py:145-169,359-378, src/models/filterbank.py:222-247."""
    from oracle import convtasnet_oracle as O
    from sepkernels import net as NET
    prev = sepkernels.set_gemm_arith("f16x3")
    try:
        K = HIP
        eps = 1e-12
        L, S, Cin = 16, 8, 1
        geo = NET.Geometry(T_in, L, S)
        F, ldt, pl = geo.F, geo.ldt, geo.pad_left
        cnt0 = N * F
        mixture = rnd(B, Cin, T_in, scale=0.3)
        E, D = rnd(N, Cin, L, scale=L ** -0.5), rnd(N, Cin, L, scale=L ** -0.5)
        g0, b0 = rnd(N, scale=0.2) + 1, rnd(N, scale=0.2)
        Wb, bb = rnd(Bn, N, 1, scale=N ** -0.5), rnd(Bn, scale=0.1)
        am = torch.tensor([0.25])
        Wm, bm = rnd(n_src * N, Sc, 1, scale=Sc ** -0.5), rnd(n_src * N, scale=0.1)
        core = padded(B, Sc, F, ldt)                                  # the TCN's skip sum
        dx0 = padded(B, Bn, F, ldt)                                   # the TCN's gradient at the bottleneck output
        d_est = rnd(B, n_src, Cin, T_in)

        def close(got, ref, tol, what):
            got, ref = got.detach().cpu().double(), ref.detach().double()
            assert torch.isfinite(got).all(), what
            err = (got - ref).abs().max().item()
            assert err <= tol * (ref.abs().max().item() + 1e-30), "{}: max err {:.3e} vs scale {:.3e}".format(what, err, ref.abs().max().item())

        dev = lambda t: to_device(t)
        f32 = dict(device=device_name(), dtype=torch.float32)
        gmix, gE, gD, gg0, gb0, gWb, gbb, gam, gWm, gbm, gcore, gdx0, gdest = (dev(t) for t in (mixture, E, D, g0, b0, Wb, bb, am, Wm, bm, core, dx0, d_est))
        pk = K.pack_weights([(gWb.view(Bn, N), Bn, N, 0), (gWb.view(Bn, N), Bn, N, 1), (gWm.view(n_src * N, Sc), n_src * N, Sc, 0), (gWm.view(n_src * N, Sc), n_src * N, Sc, 1)])
        amax = dev(torch.stack([t.abs().max() for t in (Wb, Wm)]).max().reshape(1))
        prev_amax = sepkernels.set_weights_amax(amax)
        # ---- head forward ---------------------------------------------------------------------------------------------------------------
        st0 = dev(torch.zeros(B, SLOTS, 2, dtype=torch.float64))
        w = dev(nan(B, N, ldt))
        K.encoder_fwd(gmix, gE, w, st0, B, Cin, T_in, N, L, S, F, ldt, pl, relu)
        x0 = dev(nan(B, Bn, ldt))
        K.pw_gemm(B=B, M=Bn, K=N, T=F, ldt=ldt, A=gWb, A_pk=pk[0], X=w, Y=x0, bias=gbb, pro_mode=PRO_GLN, pro_stats=st0, pro_gamma=gg0, pro_beta=gb0, count=cnt0, eps=eps)
        # ---- tail forward ---------------------------------------------------------------------------------------------------------------
        m = dev(nan(B, n_src * N, ldt))
        K.pw_gemm(B=B, M=n_src * N, K=Sc, T=F, ldt=ldt, A=gWm, A_pk=pk[2], X=gcore, Y=m, bias=gbm, pro_mode=PRO_PRELU, pro_alpha=gam, epi_flags=EPI_SIGMOID, eps=eps)
        est = dev(nan(B, n_src, Cin, T_in))
        latent = dev(nan(B, n_src, N, ldt))
        K.decoder_fwd(w, m, gD, est, latent, B, n_src, N, Cin, L, S, F, ldt, T_in, pl)
        device_sync()
        # ---- the oracle, fp64 (ReLU branch taken from the device's encoder output, see the layer test) -------------------------------------
        dd = lambda t: t.double().clone().requires_grad_(True)
        oE, oD, og0, ob0, oWb, obb, oam, oWm, obm = (dd(t) for t in (E, D, g0, b0, Wb, bb, am, Wm, bm))
        ocore = dd(core[..., :F])
        xp = torch.zeros(B, Cin, T_in + geo.padding, dtype=torch.float64)
        xp[:, :, pl:pl + T_in] = mixture.double()
        opre = O.encoder(xp, oE, S, relu=False)
        opre.retain_grad()
        w_dev = w[..., :F].cpu().double()
        if relu:
            flip = (w_dev > 0) != (opre.detach() > 0)
            assert not flip.any() or opre.detach()[flip].abs().max() <= 1e-4 * opre.detach().abs().max()
            ow = torch.where(w_dev > 0, opre, torch.zeros_like(opre))
        else:
            ow = opre
        ox0 = O.pointwise_conv(O.gln(ow, og0, ob0, eps), oWb, obb)
        opm = O.pointwise_conv(O.prelu(ocore, oam), oWm, obm)
        opm.retain_grad()
        om = 1.0 / (1.0 + torch.exp(-opm))
        olat = ow.unsqueeze(1) * om.view(B, n_src, N, F)
        oest = O.decoder(olat.reshape(B * n_src, N, F), oD, S).reshape(B, n_src, Cin, -1)[..., pl:pl + T_in]
        ((oest * d_est.double()).sum() + (ox0 * dx0[..., :F].double()).sum()).backward()
        close(w[..., :F], ow, 1e-5, "encoder")
        assert w[..., F:].abs().max().item() == 0.0
        s0 = st0.cpu().sum(1)
        assert (s0[:, 0] - ow.detach().sum((1, 2))).abs().max() <= 1e-5 * ow.detach().abs().sum((1, 2)).max()
        assert (s0[:, 1] - (ow.detach() ** 2).sum((1, 2))).abs().max() <= 1e-5 * (ow.detach() ** 2).sum((1, 2)).max()
        close(x0[..., :F], ox0, 2e-4, "bottleneck (gLN prologue)")
        close(m[..., :F], om, 2e-4, "mask (PReLU prologue, sigmoid)")
        close(latent[..., :F], olat, 2e-4, "mask * w")
        close(est, oest, 2e-4, "decoder overlap-add + crop")
        # ---- tail backward ----------------------------------------------------------------------------------------------------------------
        Fd = dev(nan(B * n_src, Cin * L, ldt))
        K.unfold(gdest, Fd, B * n_src, Cin, T_in, L, S, F, ldt, pl)
        part, _, ns = NET._wgrad(K, B, F, ldt, eps, f32, N, Cin * L, m, Fd, False, Bq=B * n_src, Gaux=w, g_mul=1, g_div=n_src)
        device_sync()
        close(part.sum(0).view(N, Cin, L), oD.grad, 5e-4, "decoder basis gradient")
        dpre, dwm = dev(nan(B, n_src * N, ldt)), dev(nan(B, N, ldt))
        K.decoder_bwd(gdest, w, m, gD, dpre, dwm, B, n_src, N, Cin, L, S, F, ldt, T_in, pl, raw_mask=0)
        device_sync()
        close(dpre[..., :F], opm.grad, 5e-4, "decoder^T: d(mask pre-activation)")
        dcore = dev(nan(B, Sc, ldt))
        dal = dev(torch.zeros(1, dtype=torch.float64))
        K.pw_gemm(B=B, M=Sc, K=n_src * N, T=F, ldt=ldt, trans_a=1, A=gWm, A_pk=pk[3], X=dpre, Y=dcore, epi_flags=EPI_PRELU_BWD, epi_aux=gcore, epi_alpha=gam, epi_dalpha=dal, eps=eps)
        device_sync()
        close(dcore[..., :F], ocore.grad, 5e-4, "mask^T with the PReLU backward")
        close(dal, oam.grad, 2e-3, "mask PReLU slope gradient")
        part, pb, ns = NET._wgrad(K, B, F, ldt, eps, f32, n_src * N, Sc, dpre, gcore, True, x_mode=PRO_PRELU, x_alpha=gam)
        device_sync()
        close(part.sum(0), oWm.grad[..., 0], 5e-4, "mask weight gradient")
        close(pb.sum(0), obm.grad, 5e-4, "mask bias gradient")
        # ---- head backward ----------------------------------------------------------------------------------------------------------------
        part, pb, ns = NET._wgrad(K, B, F, ldt, eps, f32, Bn, N, gdx0, w, True, x_mode=PRO_GLN, x_stats=st0, x_gamma=gg0, x_beta=gb0, count=cnt0)
        device_sync()
        close(part.sum(0), oWb.grad[..., 0], 5e-4, "bottleneck weight gradient")
        close(pb.sum(0), obb.grad, 5e-4, "bottleneck bias gradient")
        nt64 = ldt // 64
        dvw, rp0 = dev(nan(B, N, ldt)), dev(nan(B, N, nt64, 2))
        K.pw_gemm(B=B, M=N, K=Bn, T=F, ldt=ldt, trans_a=1, A=gWb, A_pk=pk[1], X=gdx0, Y=dvw, epi_flags=EPI_ROWSUMS, epi_aux=w, epi_rowpart=rp0, eps=eps)
        bsum0, pbeta0, pgamma0 = dev(nan(B, 2)), dev(nan(B, N)), dev(nan(B, N))
        K.gln_bwd_finalize(rp0, nt64, 2, st0, gg0, cnt0, eps, bsum0, pbeta0, pgamma0, None, B, N)
        device_sync()
        close(pgamma0.sum(0), og0.grad, 5e-4, "first gLN gain gradient")
        close(pbeta0.sum(0), ob0.grad, 5e-4, "first gLN shift gradient")
        K.head_bwd(dvw, w, dwm, st0, gg0, bsum0, B, N, F, ldt, cnt0, eps, relu)
        device_sync()
        close(dvw[..., :F], opre.grad, 5e-4, "head backward: d(encoder output) through gLN and through mask * w")
        Fx = dev(nan(B, Cin * L, ldt))
        K.unfold(gmix, Fx, B, Cin, T_in, L, S, F, ldt, pl)
        part, _, ns = NET._wgrad(K, B, F, ldt, eps, f32, N, Cin * L, dvw, Fx, False)
        device_sync()
        close(part.sum(0).view(N, Cin, L), oE.grad, 5e-4, "encoder basis gradient")
        sepkernels.set_weights_amax(prev_amax)
    finally:
        sepkernels.set_gemm_arith(prev)


@pytest.mark.parametrize("n,B,T", [(2, 4, 8000), (4, 3, 4001)])
def test_criterion_kernels_against_the_oracle(n, B, T):
    """sep_sisdr_dots / from_dots / bwd, sep_pit_search (+ sep_pit_finish) and sep_sinkhorn_fwd / bwd against the oracle's criterion functions
    (oracle.convtasnet_oracle.sisdr / neg_sisdr / pit / sinkpit) and autograd through them in fp64 -- no emulator in the chain.
    Reference: src/criterion/sdr.py:122-139,187-231, src/criterion/pit.py:9-44,163-213."""
    from oracle import convtasnet_oracle as O
    K = HIP
    eps = 1e-12
    tgt = rnd(B, n, T, scale=0.1)
    perm = torch.stack([torch.randperm(n, generator=G) for _ in range(B)])
    est = torch.gather(tgt, 1, perm.view(B, n, 1).expand(-1, -1, T)) + rnd(B, n, T, scale=0.05)      # a noisy permutation of the targets
    gest, gtgt = to_device(est), to_device(tgt)
    dots, tt, xx = (to_device(torch.zeros(*s, dtype=torch.float64)) for s in ((B, n, n), (B, n), (B, n)))
    val = to_device(nan(B, n, n))
    K.sisdr_dots(gest, gtgt, dots, tt, xx, B, n, T, True)
    K.sisdr_from_dots(dots, tt, xx, val, B, n, True, eps)
    device_sync()
    oest = est.double().clone().requires_grad_(True)
    otgt = tgt.double()
    opair = O.sisdr(oest.unsqueeze(2), otgt.unsqueeze(1), eps)                       # (B, n, n): entry [b, i, j] = SI-SDR(est_i, tgt_j)
    perr = (val.cpu().double() - opair.detach()).abs()
    assert perr[opair.detach() > -60].max() <= 1e-4 and perr.max() <= 1e-2               # dB (a pair that happens to be orthogonal to 1e-5 is ill-conditioned: its tiny dot product carries the error)
    # PIT: search + finish (loss, gradient weights, pattern) against the oracle's exhaustive search and autograd
    perms = torch.tensor(list(itertools.permutations(range(n))), dtype=torch.int32)
    P = perms.shape[0]
    best_val, best_idx = to_device(nan(B)), to_device(torch.zeros(B, dtype=torch.int64))
    gperms = to_device(perms)
    K.pit_search(val, gperms, P, n, B, True, True, best_val, best_idx)
    loss, gw, pattern = to_device(nan(1)), to_device(nan(B, n, n)), to_device(torch.zeros(B, n, dtype=torch.int64))
    K.pit_finish(best_val, best_idx, gperms, P, n, B, -1.0, 1.0 / (B * n), loss, gw, pattern)
    d_est = to_device(nan(B, n, T))
    K.sisdr_bwd(gest, gtgt, dots, tt, xx, gw, d_est, B, n, T, True, eps)
    device_sync()
    oloss, opat = O.pit(lambda a, b, batch_mean=False: O.neg_sisdr(a, b, batch_mean=batch_mean), oest, otgt)
    oloss.backward()
    assert torch.equal(pattern.cpu(), opat)
    assert torch.equal(pattern.cpu(), perm)                                            # est_i is a noisy copy of target perm[i]: the search finds it
    assert abs(loss.item() - oloss.item()) <= 1e-5 * abs(oloss.item())
    err = (d_est.cpu().double() - oest.grad).abs().max().item()
    assert err <= 1e-4 * oest.grad.abs().max().item(), err
    # SinkPIT on the same pair matrix (NegSI-SDR costs): loss per item, soft assignment, gradient with respect to the costs
    for iters, cold in ((10, 1.0), (50, 0.5)):
        C = (-val).contiguous()
        zwork = to_device(torch.zeros(B, 2 * iters + 1, n, n, dtype=torch.float64))
        sl, sP, dC = to_device(nan(B)), to_device(nan(B, n, n)), to_device(nan(B, n, n))
        K.sinkhorn_fwd(C, zwork, sl, sP, B, n, cold, iters)
        dl = rnd(B)
        K.sinkhorn_bwd(C, zwork, to_device(dl), dC, B, n, cold, iters)
        device_sync()
        oC = (-opair.detach()).clone().requires_grad_(True)
        pair = lambda xi, tj, batch_mean=False: oC.reshape(-1)                          # the oracle's sinkpit evaluates pairs in (b, i, j) order
        osl, oP = O.sinkpit(pair, oest.detach(), otgt, coldness=cold, iteration=iters, batch_mean=False)
        (osl * dl.double()).sum().backward()
        assert (sl.cpu().double() - osl.detach()).abs().max() <= 1e-4 * osl.detach().abs().max()
        assert (sP.cpu().double() - oP.detach()).abs().max() <= 1e-4
        assert (dC.cpu().double() - oC.grad).abs().max() <= 2e-4 * oC.grad.abs().max()


@pytest.mark.parametrize("n", [1, 1000, 4984881])
def test_absmax_and_memset(n):
    """sep_absmax (the A-operand bound of the split arithmetic over the flat parameter buffer in a recorded step) and sep_memset"""
    x = rnd(n)
    x[n // 2] = -7.5 if n > 1 else 0.25
    gx, out = to_device(x), to_device(nan(1))
    HIP.absmax(gx, out, n)
    device_sync()
    assert out.cpu().item() == x.abs().max().item()
    buf = to_device(rnd(n))
    HIP.memset(buf, 0)
    device_sync()
    assert buf.cpu().abs().max().item() == 0.0
    z = HIP.zeros(3, 5, device=device_name(), dtype=torch.float64)
    assert z.shape == (3, 5) and z.dtype == torch.float64 and z.abs().max().item() == 0.0
