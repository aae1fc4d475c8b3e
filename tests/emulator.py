"""
TEST INFRASTRUCTURE ONLY -- a CPU emulation of every entry point of include/sepkernels.h with plain torch
arithmetic, operating on the same padded (batch, channel, ldt) buffers and honouring the same contracts
(zeroed pad frames, partial-slab outputs, accumulated statistics).

Purpose: (1) check the host-side orchestration (sepkernels/net.py, the autograd wrappers, the criteria) against
the oracle in this GPU-less container; (2) serve as the per-kernel expected value in the `-m gpu` tests, where
each HIP kernel is compared against its emulation on identical buffers.

It is injected with sepkernels._set_backend_for_tests() by tests only; the product never imports it.
"""
import math

import torch

import sepkernels

PRO_NONE, PRO_PRELU, PRO_GLN, PRO_GLN_PRELU, PRO_GLN_BWD = 0, 1, 2, 3, 4
EPI_STATS_PRELU, EPI_RESIDUAL, EPI_SIGMOID, EPI_PRELU_BWD, EPI_ROWSUMS, EPI_ROWSUMS_PRELU = 1, 2, 4, 8, 16, 32


def _prelu(x, a):
    return torch.where(x > 0, x, a.reshape(()).to(x.dtype) * x)


def _prelu_grad(x, a):
    return torch.where(x > 0, torch.ones_like(x), a.reshape(()).to(x.dtype) * torch.ones_like(x))


SLOTS = 16


def _tot(stats):
    """(B, SLOTS, 2) slotted accumulators -> (B, 2) totals"""
    return stats.reshape(-1, SLOTS, 2).sum(1)


def _acc(stats, s, ss):
    v = stats.reshape(-1, SLOTS, 2)
    v[:, 0, 0] += s.double()
    v[:, 0, 1] += ss.double()


def _mu_rstd(stats, count, eps, dtype):
    st = _tot(stats)
    m = st[:, 0] / count
    var = (st[:, 1] / count - m * m).clamp_min(0.0)
    return m.to(dtype).view(-1, 1, 1), (1.0 / torch.sqrt(var + eps)).to(dtype).view(-1, 1, 1)


def _publish(bacc, stats, arrive, bsum, arrivals, expected, count, eps):
    """gln_bwd_publish of csrc/common.hpp: the producers' arrivals are counted per sample; the call that completes `expected` turns the
    slotted sums {sum_c gamma_c sum_t g, sum_c gamma_c sum_t g u} and the gLN's statistics into the two means (fp64 arithmetic)"""
    top = arrive.reshape(-1, 17)[:, 16]          # the emulator only keeps the samples' top-level counters (one arrival per call)
    top += arrivals
    if int(top[0]) < expected:
        return
    assert int(top.max()) == expected and int(top.min()) == expected, "arrival counters out of step"
    st, ba = _tot(stats), _tot(bacc)
    m = st[:, 0] / count
    var = (st[:, 1] / count - m * m).clamp_min(0.0)
    r = 1.0 / torch.sqrt(var + eps)
    bs = bsum.reshape(-1, 2)
    bs[:, 0] = (ba[:, 0] / count).to(bs.dtype)
    bs[:, 1] = (r * (ba[:, 1] - m * ba[:, 0]) / count).to(bs.dtype)


class EmuBackend:
    name = "emulator"

    # ------------------------------------------------------------------ GEMMs
    def pack_weights(self, specs):
        """No arithmetic here: the emulator multiplies with the fp32 weights; the PackedA only remembers its source so that
        pw_gemm can check that the host code pairs every product with the right pack."""
        for W, r, c, t in specs:            # the same refusals as sep_pack_weights (csrc/gemm_coop.hip): found the hard way, see
            M, K = (c if t else r), (r if t else c)      # test_modules_cpu.py::test_widths_in_odd_multiples_of_16
            if K % 16 or M % 32:
                raise sepkernels.SepKernelsError("sep_pack_weights: bad segment (M={} K={})".format(M, K))
        return [sepkernels.PackedA(None, None, (c if t else r), (r if t else c), src=(W, r, c, t)) for W, r, c, t in specs]

    def pw_gemm(self, *, B, M, K, T, ldt, A, X, Y, trans_a=0, A2=None, X2=None, k_split=0, Y2=None, m_split=0,
                pro_mode=PRO_NONE, epi_flags=0, accumulate=0, eps=1e-12, count=0.0, bias=None, pro_alpha=None,
                pro_stats=None, pro_gamma=None, pro_beta=None, pro_aux=None, pro_bsum=None, pro_bacc=None, pro_store=None,
                pro_dalpha=None, epi_alpha=None, epi_stats=None, epi_res=None, epi_aux=None, epi_dalpha=None,
                epi_rowpart=None, arith=None, a_amax=None, A_pk=None):
        dt = X.dtype
        k1 = k_split if k_split else K
        if trans_a:
            Am = A.reshape(k1, M)
            if k_split:
                Am = torch.cat([Am, A2.reshape(K - k1, M)], 0)
            Am = Am.t()
        else:
            Am = A.reshape(M, k1)
            if k_split:
                Am = torch.cat([Am, A2.reshape(M, K - k1)], 1)
        if A_pk is not None:
            # plumbing check of the packed-weight path: the pack handed over must BE the matrix of this product
            W, r, c, t = A_pk.src
            Apk = W.reshape(-1)[:r * c].reshape(r, c)
            Apk = Apk.t() if t else Apk
            assert (A_pk.M, A_pk.K) == (M, K) and tuple(Apk.shape) == (M, K) and torch.equal(Apk, Am), "wrong packed weights for this product"
        Xf = X.reshape(B, k1, ldt)
        if k_split:
            Xf = torch.cat([Xf, X2.reshape(B, K - k1, ldt)], 1)
        valid = (torch.arange(ldt) < T).view(1, 1, ldt)
        if pro_mode in (PRO_GLN, PRO_GLN_PRELU, PRO_GLN_BWD):
            mu, rstd = _mu_rstd(pro_stats, count, eps, dt)
        if pro_mode == PRO_PRELU:
            Xp = _prelu(Xf, pro_alpha)
        elif pro_mode in (PRO_GLN, PRO_GLN_PRELU):
            u = _prelu(Xf, pro_alpha) if pro_mode == PRO_GLN_PRELU else Xf
            sc = pro_gamma.view(1, K, 1) * rstd
            sh = pro_beta.view(1, K, 1) - mu * sc
            Xp = u * sc + sh
        elif pro_mode == PRO_GLN_BWD:
            if pro_store.data_ptr() == X.data_ptr() and M > 128:       # the library's refusal (sep_pw_gemm): in place only with one row tile
                raise RuntimeError("sep_pw_gemm: pro_store may alias X only when M <= 128 (one row tile)")
            a = pro_aux.reshape(B, K, ldt)
            u = _prelu(a, pro_alpha)
            xh = (u - mu) * rstd
            if pro_bacc is not None:        # gln_bwd_means: the kernel forms the two means from the producer's slots
                st, ba = _tot(pro_stats), _tot(pro_bacc)
                m64 = st[:, 0] / count
                r64 = 1.0 / torch.sqrt((st[:, 1] / count - m64 * m64).clamp_min(0.0) + eps)
                mg, mgx = (ba[:, 0] / count).to(dt).view(B, 1, 1), (r64 * (ba[:, 1] - m64 * ba[:, 0]) / count).to(dt).view(B, 1, 1)
            else:
                mg, mgx = pro_bsum[:, 0].view(B, 1, 1), pro_bsum[:, 1].view(B, 1, 1)
            du = rstd * (pro_gamma.view(1, K, 1) * Xf - mg - xh * mgx)
            da = torch.where(valid, du * _prelu_grad(a, pro_alpha), torch.zeros_like(du))
            pro_dalpha += torch.where(valid & (a <= 0), du * a, torch.zeros_like(du)).sum().double()
            Xp = da
            pro_store.reshape(B, K, ldt).copy_(da)
        else:
            Xp = Xf
        y = torch.einsum("mk,bkt->bmt", Am, Xp)
        if bias is not None:
            y = y + bias.reshape(1, M, 1)
        Mf = m_split if m_split else M
        if epi_flags & EPI_STATS_PRELU:
            u = torch.where(valid, _prelu(y, epi_alpha), torch.zeros_like(y))
            _acc(epi_stats, u.sum((1, 2)), (u * u).sum((1, 2)))
        if epi_flags & EPI_RESIDUAL:
            y = torch.cat([y[:, :Mf] + epi_res.reshape(B, -1, ldt)[:, :Mf], y[:, Mf:]], 1)
        if epi_flags & EPI_SIGMOID:
            y = 1.0 / (1.0 + torch.exp(-y))
        if epi_flags & EPI_PRELU_BWD:
            s = epi_aux.reshape(B, M, ldt)
            epi_dalpha += torch.where(valid & (s <= 0), y * s, torch.zeros_like(y)).sum().double()
            y = y * _prelu_grad(s, epi_alpha)
        if epi_flags & EPI_ROWSUMS:
            u = epi_aux.reshape(B, M, ldt)
            if epi_flags & EPI_ROWSUMS_PRELU:
                u = _prelu(u, epi_alpha)
            yv = torch.where(valid, y, torch.zeros_like(y))
            rp = epi_rowpart.reshape(B, M, ldt // 64, 2)
            rp[..., 0] = yv.reshape(B, M, ldt // 64, 64).sum(-1)
            rp[..., 1] = (yv * u).reshape(B, M, ldt // 64, 64).sum(-1)
        y = torch.where(valid, y, torch.zeros_like(y))
        if m_split:
            Y.reshape(B, Mf, ldt).copy_(y[:, :Mf])
            y2 = Y2.reshape(B, M - Mf, ldt)
            y2.copy_(y2 + y[:, Mf:] if accumulate else y[:, Mf:])
        else:
            yv = Y.reshape(B, M, ldt)
            yv.copy_(yv + y if accumulate else y)

    def pw_wgrad(self, *, B, M, N, T, ldt, G, X, partial, nsplit, G2=None, g_split=0, Gaux=None, g_mul=0, g_div=1,
                 x_mode=PRO_NONE, x_div=1, eps=1e-12, count=0.0, x_alpha=None, x_stats=None, x_gamma=None, x_beta=None,
                 partial_bias=None, arith=None):
        dt = G.dtype
        m1 = g_split if g_split else M
        Gf = G.reshape(B, m1, ldt)
        if g_split:
            Gf = torch.cat([Gf, G2.reshape(B, M - m1, ldt)], 1)
        if g_mul:
            Gf = Gf * Gaux.reshape(B // g_div, M, ldt).repeat_interleave(g_div, 0)
        Xf = X.reshape(B // x_div, N, ldt).repeat_interleave(x_div, 0)
        if x_mode == PRO_PRELU:
            Xp = _prelu(Xf, x_alpha)
        elif x_mode in (PRO_GLN, PRO_GLN_PRELU):
            mu, rstd = _mu_rstd(x_stats, count, eps, dt)
            mu, rstd = mu.repeat_interleave(x_div, 0), rstd.repeat_interleave(x_div, 0)
            u = _prelu(Xf, x_alpha) if x_mode == PRO_GLN_PRELU else Xf
            sc = x_gamma.view(1, N, 1) * rstd
            Xp = u * sc + (x_beta.view(1, N, 1) - mu * sc)
        else:
            Xp = Xf
        partial.zero_()
        if partial_bias is not None:
            partial_bias.zero_()
        k = nsplit // B if nsplit % B == 0 else 0
        if k and (ldt // 32) % k == 0:
            # sample-aligned slabs (include/sepkernels.h, sep_gln_bwd_from_wgrad): slab b * k + j holds frames [j ldt / k, (j + 1) ldt / k) of sample b
            partial.reshape(B, k, M, N).copy_(torch.einsum("bmkt,bnkt->bkmn", Gf.reshape(B, M, k, ldt // k), Xp.reshape(B, N, k, ldt // k)))
            if partial_bias is not None:
                partial_bias.reshape(B, k, M).copy_(Gf.reshape(B, M, k, ldt // k).sum(3).permute(0, 2, 1))
        else:       # how the frames are spread over the slabs is the kernel's business: everything in slab 0
            partial.reshape(nsplit, M, N)[0] = torch.einsum("bmt,bnt->mn", Gf, Xp)
            if partial_bias is not None:
                partial_bias.reshape(nsplit, M)[0] = Gf.sum((0, 2))

    def reduce_slabs(self, segs):
        for (src, off, dst, n, nslab, stride, acc, scale) in segs:
            flat = src.reshape(-1)
            tot = torch.zeros(n, dtype=src.dtype)
            for s in range(nslab):
                tot += flat[off + s * stride: off + s * stride + n]
            tot = tot * scale
            d = dst.reshape(-1)
            d.copy_(d + tot if acc else tot)

    def f64_to_f32(self, src, dst, n, accumulate=0):
        d = dst.reshape(-1)
        v = src.reshape(-1)[:n].to(dst.dtype)
        d[:n] = d[:n] + v if accumulate else v

    # ------------------------------------------------------------------ encoder / decoder
    @staticmethod
    def _xpad(x, pad_left, total):
        Bp, C, Tin = x.shape
        xp = torch.zeros(Bp, C, total, dtype=x.dtype)
        xp[:, :, pad_left:pad_left + Tin] = x
        return xp

    def unfold(self, x, frames, Bp, C, Tin, L, S, F, ldt, pad_left):
        xp = self._xpad(x.reshape(Bp, C, Tin), pad_left, S * (F - 1) + L + S)
        idx = (torch.arange(F) * S).view(1, F) + torch.arange(L).view(L, 1)      # (L, F)
        fr = xp[:, :, idx]                                                       # (Bp, C, L, F)
        out = frames.reshape(Bp, C * L, ldt)
        out.zero_()
        out[:, :, :F] = fr.reshape(Bp, C * L, F)

    def encoder_fwd(self, x, E, w, stats, B, Cin, Tin, N, L, S, F, ldt, pad_left, relu):
        fr = torch.zeros(B, Cin * L, ldt, dtype=x.dtype)
        self.unfold(x, fr, B, Cin, Tin, L, S, F, ldt, pad_left)
        y = torch.einsum("nq,bqf->bnf", E.reshape(N, Cin * L), fr)
        if relu:
            y = y.clamp_min(0)
        y[:, :, F:] = 0
        w.reshape(B, N, ldt).copy_(y)
        _acc(stats, y.sum((1, 2)), (y * y).sum((1, 2)))

    def decoder_fwd(self, w, m, D, est, latent, B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left):
        wh = w.reshape(B, 1, N, ldt) * m.reshape(B, n_src, N, ldt)
        wh[..., F:] = 0
        if latent is not None:
            latent.reshape(B, n_src, N, ldt).copy_(wh)
        fr = torch.einsum("bsnf,nck->bsckf", wh[..., :F], D.reshape(N, Cout, L))
        total = S * (F - 1) + L
        out = torch.zeros(B, n_src, Cout, total, dtype=w.dtype)
        idx = ((torch.arange(F) * S).view(1, F) + torch.arange(L).view(L, 1)).reshape(-1)
        out.index_add_(3, idx, fr.reshape(B, n_src, Cout, L * F))
        est.reshape(B, n_src, Cout, Tout).copy_(out[..., pad_left:pad_left + Tout])

    def softmax_ch_fwd(self, y, B, C, T, ldt):
        v = y.reshape(B, C, ldt)
        out = torch.zeros_like(v)
        out[..., :T] = torch.softmax(v[..., :T], dim=1)
        v.copy_(out)

    def softmax_ch_bwd(self, y, g, B, C, T, ldt):
        yv, gv = y.reshape(B, C, ldt), g.reshape(B, C, ldt)
        out = yv * (gv - (gv * yv).sum(1, keepdim=True))
        out[..., T:] = 0
        gv.copy_(out)

    def decoder_bwd(self, d_est, w, m, D, dpre, dwm, B, n_src, N, Cout, L, S, F, ldt, Tout, pad_left, raw_mask=0):
        fr = torch.zeros(B * n_src, Cout * L, ldt, dtype=w.dtype)
        self.unfold(d_est.reshape(B * n_src, Cout, Tout), fr, B * n_src, Cout, Tout, L, S, F, ldt, pad_left)
        dl = torch.einsum("nq,bqf->bnf", D.reshape(N, Cout * L), fr).reshape(B, n_src, N, ldt)
        mv = m.reshape(B, n_src, N, ldt)
        wv = w.reshape(B, 1, N, ldt)
        dp = dl * wv if raw_mask else dl * wv * mv * (1 - mv)
        dp[..., F:] = 0
        dpre.reshape(B, n_src, N, ldt).copy_(dp)
        dw = (dl * mv).sum(1)
        dw[..., F:] = 0
        dwm.reshape(B, N, ldt).copy_(dw)

    # ------------------------------------------------------------------ depthwise
    def dwconv_fwd(self, a, stats1, gamma1, beta1, alpha1, wd, bd, alpha2, z, stats2, B, C, T, ldt, dilation, eps):
        dt = a.dtype
        mu, rstd = _mu_rstd(stats1, float(C * T), eps, dt)
        sc = gamma1.view(1, C, 1) * rstd
        v = _prelu(a.reshape(B, C, ldt), alpha1) * sc + (beta1.view(1, C, 1) - mu * sc)
        d = dilation
        vp = torch.zeros(B, C, T + 2 * d, dtype=dt)
        vp[:, :, d:d + T] = v[:, :, :T]
        wk = wd.reshape(C, 3)
        zz = bd.view(1, C, 1) + wk[:, 0].view(1, C, 1) * vp[:, :, 0:T] + wk[:, 1].view(1, C, 1) * vp[:, :, d:d + T] \
            + wk[:, 2].view(1, C, 1) * vp[:, :, 2 * d:2 * d + T]
        out = z.reshape(B, C, ldt)
        out.zero_()
        out[:, :, :T] = zz
        u = _prelu(zz, alpha2)
        _acc(stats2, u.sum((1, 2)), (u * u).sum((1, 2)))

    def dwconv_bwd(self, dv2, z, a, stats1, gamma1, beta1, alpha1, stats2, gamma2, alpha2, bsum2, wd, bd, dv1, rowpart, bacc1, arrive1, bsum1,
                   B, C, T, ldt, dilation, eps):
        dt = a.dtype
        d = dilation
        cnt = float(C * T)
        mu1, r1 = _mu_rstd(stats1, cnt, eps, dt)
        mu2, r2 = _mu_rstd(stats2, cnt, eps, dt)
        zz = z.reshape(B, C, ldt)[:, :, :T]
        aa = a.reshape(B, C, ldt)[:, :, :T]
        g = dv2.reshape(B, C, ldt)[:, :, :T]
        u2 = _prelu(zz, alpha2)
        xh = (u2 - mu2) * r2
        du2 = r2 * (gamma2.view(1, C, 1) * g - bsum2[:, 0].view(B, 1, 1) - xh * bsum2[:, 1].view(B, 1, 1))
        dz = du2 * _prelu_grad(zz, alpha2)
        dal = torch.where(zz <= 0, du2 * zz, torch.zeros_like(zz)).sum(2)
        u1 = _prelu(aa, alpha1)
        sc1 = gamma1.view(1, C, 1) * r1
        v1 = u1 * sc1 + (beta1.view(1, C, 1) - mu1 * sc1)
        dzp = torch.zeros(B, C, T + 2 * d, dtype=dt)
        dzp[:, :, d:d + T] = dz
        v1p = torch.zeros(B, C, T + 2 * d, dtype=dt)
        v1p[:, :, d:d + T] = v1
        wk = wd.reshape(C, 3)
        # dv1[t] = w0 dz[t+d] + w1 dz[t] + w2 dz[t-d]
        dv = wk[:, 0].view(1, C, 1) * dzp[:, :, 2 * d:2 * d + T] + wk[:, 1].view(1, C, 1) * dz + wk[:, 2].view(1, C, 1) * dzp[:, :, 0:T]
        out = dv1.reshape(B, C, ldt)
        out.zero_()
        out[:, :, :T] = dv
        ntile = (ldt + 1023) // 1024
        rp = rowpart.reshape(B, C, ntile, 8)
        rp.zero_()
        rp[:, :, 0, 0] = dv.sum(2)
        rp[:, :, 0, 1] = (dv * u1).sum(2)
        rp[:, :, 0, 2] = dz.sum(2)
        rp[:, :, 0, 3] = (dz * v1p[:, :, 0:T]).sum(2)
        rp[:, :, 0, 4] = (dz * v1).sum(2)
        rp[:, :, 0, 5] = (dz * v1p[:, :, 2 * d:2 * d + T]).sum(2)
        rp[:, :, 0, 6] = dal
        if bacc1 is not None:
            g1 = gamma1.view(1, C, 1).to(dt)
            _acc(bacc1, (g1 * dv).sum((1, 2)), (g1 * dv * u1).sum((1, 2)))
            if arrive1 is not None:
                _publish(bacc1, stats1, arrive1, bsum1, 1, 1, cnt, eps)

    def gln_bwd_finalize(self, rowpart, ntile, nq, stats, gamma, count, eps, bsum, pbeta, pgamma, pextra, B, C):
        dt = rowpart.dtype
        rp = rowpart.reshape(B, C, ntile, nq).sum(2)
        mu, rstd = _mu_rstd(stats, count, eps, dt)
        R1, R2 = rp[..., 0], rp[..., 1]
        pg = rstd.view(B, 1) * (R2 - mu.view(B, 1) * R1)
        pbeta.reshape(B, C).copy_(R1)
        pgamma.reshape(B, C).copy_(pg)
        if bsum is not None:
            bsum.reshape(B, 2)[:, 0] = (gamma.view(1, C) * R1).sum(1) / count
            bsum.reshape(B, 2)[:, 1] = (gamma.view(1, C) * pg).sum(1) / count
        if nq == 8:
            pe = pextra.reshape(-1)
            slab = pe[:B * 4 * C].reshape(B, 4 * C)
            slab[:, :C] = rp[..., 2]
            slab[:, C:] = rp[..., 3:6].reshape(B, 3 * C)
            pe[B * 4 * C:B * 4 * C + B] = rp[..., 6].sum(1)
            pe[B * 4 * C + B:B * 4 * C + B + B * C] = rp[..., 6].reshape(-1)      # per-row scratch of the two-kernel finalize

    def gln_bwd_finalize_batch(self, segs):
        for sg in segs:
            self.gln_bwd_finalize(*sg)

    def gln_bwd_from_wgrad(self, part, part_bias, W, stats, gamma, beta, count, eps, dW_b, pbeta, pgamma, bacc, arrive, bsum, B, M, N,
                           slabs_per_sample, accumulate=0, products=1):
        dt = part.dtype
        raw = part.reshape(B, slabs_per_sample, M, N).sum(1)
        gs = part_bias.reshape(B, slabs_per_sample, M).sum(1)
        Wm = W.reshape(-1)[:M * N].reshape(M, N).to(dt)
        mu, rstd = _mu_rstd(stats, count, eps, dt)
        mu, rstd = mu.view(B, 1), rstd.view(B, 1)
        R1 = torch.einsum("mn,bm->bn", Wm, gs)
        R2 = (Wm.unsqueeze(0) * raw).sum(1)
        sc = gamma.view(1, N) * rstd
        sh = beta.view(1, N) - mu * sc
        dW_b.reshape(B, M, N).copy_(sc.unsqueeze(1) * raw + sh.unsqueeze(1) * gs.unsqueeze(2))
        pb, pg = pbeta.reshape(B, N), pgamma.reshape(B, N)
        if accumulate:
            pb += R1
            pg += rstd * (R2 - mu * R1)
        else:
            pb.copy_(R1)
            pg.copy_(rstd * (R2 - mu * R1))
        _acc(bacc, (gamma.view(1, N) * R1).sum(1), (gamma.view(1, N) * R2).sum(1))
        _publish(bacc, stats, arrive, bsum, 1, products, count, eps)

    def head_bwd(self, dvw, w, dwm, stats0, gamma0, bsum0, B, C, T, ldt, count, eps, relu):
        dt = w.dtype
        mu, rstd = _mu_rstd(stats0, count, eps, dt)
        g = dvw.reshape(B, C, ldt)
        wv = w.reshape(B, C, ldt)
        xh = (wv - mu) * rstd
        v = rstd * (gamma0.view(1, C, 1) * g - bsum0[:, 0].view(B, 1, 1) - xh * bsum0[:, 1].view(B, 1, 1)) + dwm.reshape(B, C, ldt)
        if relu:
            v = torch.where(wv > 0, v, torch.zeros_like(v))
        v[:, :, T:] = 0
        g.copy_(v)

    # ------------------------------------------------------------------ cumulative layer norm
    def cln_ws_bytes(self, B, C, T, ldt):
        return B * 2 * ldt * 8

    def cln_fwd(self, x, gamma, beta, y, mean, rstd, ws, B, C, T, ldt, eps, alpha=None):
        if alpha is not None:
            x = _prelu(x, alpha)
        v = x.reshape(B, C, ldt)[:, :, :T].double()
        n = torch.arange(1, T + 1, dtype=torch.float64) * C
        m = v.sum(1).cumsum(1) / n
        var = ((v * v).sum(1).cumsum(1) / n - m * m).clamp_min(0)
        r = 1.0 / (var.sqrt() + eps)
        mean.reshape(B, ldt)[:, :T] = m.to(mean.dtype)           # rows of ldt; fp32 on the device; the fp64 emulator runs of the CPU tests keep fp64
        rstd.reshape(B, ldt)[:, :T] = r.to(rstd.dtype)
        out = torch.zeros(B, C, ldt, dtype=x.dtype)
        mf, rf = mean.reshape(B, 1, ldt)[:, :, :T], rstd.reshape(B, 1, ldt)[:, :, :T]
        out[:, :, :T] = (x.reshape(B, C, ldt)[:, :, :T] - mf) * rf * gamma.view(1, C, 1) + beta.view(1, C, 1)
        y.reshape(B, C, ldt).copy_(out)

    def cln_bwd(self, dy, x, gamma, mean, rstd, dx, dgamma_part, dbeta_part, ws, B, C, T, ldt, eps, alpha=None, dalpha_part=None):
        g = dy.reshape(B, C, ldt)[:, :, :T].double()
        x_pre = x.reshape(B, C, ldt)[:, :, :T].double()
        v = _prelu(x_pre, alpha) if alpha is not None else x_pre
        m, r = mean.reshape(B, 1, ldt)[:, :, :T].double(), rstd.reshape(B, 1, ldt)[:, :, :T].double()
        gh = g * gamma.view(1, C, 1).double()
        A, Bq = gh.sum(1), (gh * (v - m)).sum(1)
        n = torch.arange(1, T + 1, dtype=torch.float64) * C
        sigma = 1.0 / r[:, 0] - eps
        Dq = torch.where(sigma > 0, -Bq * r[:, 0] ** 2 / (2 * sigma.clamp_min(1e-300)), torch.zeros_like(Bq))
        Dm = -r[:, 0] * A - 2 * m[:, 0] * Dq
        P = (Dm / n).flip(1).cumsum(1).flip(1).unsqueeze(1)
        Q = (Dq / n).flip(1).cumsum(1).flip(1).unsqueeze(1)
        out = torch.zeros(B, C, ldt, dtype=x.dtype)
        du = gh * r + P + 2 * v * Q
        if alpha is not None:
            dalpha_part.reshape(B, C).copy_(torch.where(x_pre <= 0, du * x_pre, torch.zeros_like(du)).sum(2).to(dalpha_part.dtype))
            du = du * _prelu_grad(x_pre, alpha)
        out[:, :, :T] = du.to(x.dtype)
        dx.reshape(B, C, ldt).copy_(out)
        dgamma_part.reshape(B, C).copy_((g * (v - m) * r).sum(2).to(dgamma_part.dtype))
        dbeta_part.reshape(B, C).copy_(g.sum(2).to(dbeta_part.dtype))

    # ------------------------------------------------------------------ attention core (csrc/attn.hip)
    @staticmethod
    def _attn_keep(N, L, H, p_drop, seed):
        """the dropout decision of csrc/attn.hip: a hash of (seed, n, h, query, key), 32-bit arithmetic"""
        if not p_drop > 0:
            return None, 1.0
        M = 0xFFFFFFFF
        thr = max(1, min(int(p_drop * 4294967296.0), M))
        s0, s1 = seed & M, (seed >> 32) & M
        n, h, q, k = torch.meshgrid(torch.arange(N), torch.arange(H), torch.arange(L), torch.arange(L), indexing="ij")
        idx = ((n * H + h) * L + q) * L + k                                             # 64-bit element index: the high word enters the hash too
        x = (idx & M) ^ s0
        x = (x * 0x9E3779B1) & M
        x = x ^ (x >> 15)
        x = (x * 0x85EBCA6B) & M
        x = x ^ (x >> 13)
        x = (x + s1 + (((idx >> 32) & M) * 0x9E3779B1)) & M
        x = (x * 0xC2B2AE35) & M
        x = x ^ (x >> 16)
        return x >= thr, 1.0 / (1.0 - p_drop)

    def attn_fwd(self, qkv, o, lse, N, L, H, D, scale, p_drop=0.0, seed=0):
        t = qkv.reshape(N, L, 3, H, D).double()
        q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))                   # (N, H, L, D)
        s = scale * q @ k.transpose(-1, -2)
        ls = torch.logsumexp(s, dim=-1)
        pr = torch.exp(s - ls.unsqueeze(-1))
        keep, kinv = self._attn_keep(N, L, H, p_drop, seed)
        pd = pr if keep is None else torch.where(keep, pr * kinv, torch.zeros_like(pr))
        o.reshape(N, L, H, D).copy_((pd @ v).permute(0, 2, 1, 3).to(o.dtype))
        lse.reshape(N, H, L).copy_(ls.to(lse.dtype))

    def attn_bwd(self, qkv, o, dout, lse, delta, dqkv, N, L, H, D, scale, p_drop=0.0, seed=0):
        t = qkv.reshape(N, L, 3, H, D).double()
        q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        g = dout.reshape(N, L, H, D).double().permute(0, 2, 1, 3)
        oo = o.reshape(N, L, H, D).double().permute(0, 2, 1, 3)
        pr = torch.exp(scale * q @ k.transpose(-1, -2) - lse.reshape(N, H, L, 1).double())
        keep, kinv = self._attn_keep(N, L, H, p_drop, seed)
        dl = (g * oo).sum(-1, keepdim=True)
        dpd = g @ v.transpose(-1, -2)
        if keep is None:
            pd, dp = pr, dpd
        else:
            pd, dp = torch.where(keep, pr * kinv, torch.zeros_like(pr)), torch.where(keep, dpd * kinv, torch.zeros_like(dpd))
        ds = pr * (dp - dl)
        out = torch.stack([scale * ds @ k, scale * ds.transpose(-1, -2) @ q, pd.transpose(-1, -2) @ g], dim=0)      # (3, N, H, L, D)
        dqkv.reshape(N, L, 3, H, D).copy_(out.permute(1, 3, 0, 2, 4).to(dqkv.dtype))
        delta.reshape(N, H, L).copy_(dl[..., 0].to(delta.dtype))

    # ------------------------------------------------------------------ gLN on token-major rows
    def gln_tokens_ws_bytes(self, nseq, L, C):
        return 0

    def gln_tokens_fwd(self, x, gamma, beta, y, stats, nseq, L, C, eps, ws=None):
        v = x.reshape(nseq, L * C).double()
        m = v.mean(1, keepdim=True)
        r = 1.0 / torch.sqrt((v * v).mean(1, keepdim=True) - m * m + eps)
        stats.reshape(nseq, 2)[:, 0] = m[:, 0].to(stats.dtype)
        stats.reshape(nseq, 2)[:, 1] = r[:, 0].to(stats.dtype)
        mu, rs = stats.reshape(nseq, 2)[:, 0].view(nseq, 1, 1), stats.reshape(nseq, 2)[:, 1].view(nseq, 1, 1)
        y.reshape(nseq, L, C).copy_((x.reshape(nseq, L, C) - mu) * rs * gamma.view(1, 1, C) + beta.view(1, 1, C))

    def gln_tokens_bwd(self, dy, x, gamma, stats, dx, part, nseq, L, C, ws=None):
        mu, rs = stats.reshape(nseq, 2)[:, 0].view(nseq, 1, 1), stats.reshape(nseq, 2)[:, 1].view(nseq, 1, 1)
        g = dy.reshape(nseq, L, C)
        xh = (x.reshape(nseq, L, C) - mu) * rs
        gg = g * gamma.view(1, 1, C)
        m1, m2 = gg.double().mean((1, 2), keepdim=True).to(g.dtype), (gg * xh).double().mean((1, 2), keepdim=True).to(g.dtype)
        dx.reshape(nseq, L, C).copy_(rs * (gg - m1 - xh * m2))
        p = part.reshape(nseq, 2, C)
        p[:, 0] = (g * xh).sum(1)
        p[:, 1] = g.sum(1)

    # ------------------------------------------------------------------ residual sum + layer norm over the features of rows (csrc/rownorm.hip)
    @staticmethod
    def _rownorm_keep(n, p_drop, seed):
        """the dropout decision of csrc/rownorm.hip: the hash of csrc/attn.hip on the flat element index"""
        if not p_drop > 0:
            return None, 1.0
        M = 0xFFFFFFFF
        thr = max(1, min(int(p_drop * 4294967296.0), M))
        s0, s1 = seed & M, (seed >> 32) & M
        idx = torch.arange(n)
        x = (idx & M) ^ s0
        x = (x * 0x9E3779B1) & M
        x = x ^ (x >> 15)
        x = (x * 0x85EBCA6B) & M
        x = x ^ (x >> 13)
        x = (x + s1 + (((idx >> 32) & M) * 0x9E3779B1)) & M
        x = (x * 0xC2B2AE35) & M
        x = x ^ (x >> 16)
        return x >= thr, 1.0 / (1.0 - p_drop)

    def rownorm_parts(self, rows, C):
        return min((rows + 3) // 4, 2048)

    def rownorm_fwd(self, x, res, gamma, beta, s, y, stat, rows, C, eps, p_drop=0.0, seed=0):
        assert (res is not None or s is None) and (p_drop == 0 or res is not None)
        v = x.reshape(rows, C)
        if res is not None:
            q = res.reshape(rows, C)
            keep, kinv = self._rownorm_keep(rows * C, p_drop, seed)
            if keep is not None:
                q = torch.where(keep.view(rows, C).to(q.device), q * kinv, torch.zeros_like(q))
            v = v + q
            if s is not None:
                s.reshape(rows, C).copy_(v)
        d = v.double()
        m = d.mean(1, keepdim=True)
        r = 1.0 / torch.sqrt(((d - m) ** 2).mean(1, keepdim=True) + eps)
        st = stat.reshape(rows, 2)
        st[:, 0] = m[:, 0].to(st.dtype)
        st[:, 1] = r[:, 0].to(st.dtype)
        y.reshape(rows, C).copy_((v - st[:, :1]) * st[:, 1:] * gamma.view(1, C) + beta.view(1, C))

    def rownorm_bwd(self, dy, s, gamma, stat, ds, dres, part, rows, C, p_drop=0.0, seed=0):
        assert (p_drop > 0) == (dres is not None)
        st = stat.reshape(rows, 2)
        g = dy.reshape(rows, C)
        xh = (s.reshape(rows, C) - st[:, :1]) * st[:, 1:]
        gg = g * gamma.view(1, C)
        m1, m2 = gg.double().mean(1, keepdim=True).to(g.dtype), (gg * xh).double().mean(1, keepdim=True).to(g.dtype)
        o = st[:, 1:] * (gg - m1 - xh * m2)
        ds.reshape(rows, C).copy_(o)
        if dres is not None:
            keep, kinv = self._rownorm_keep(rows * C, p_drop, seed)
            dres.reshape(rows, C).copy_(torch.where(keep.view(rows, C).to(o.device), o * kinv, torch.zeros_like(o)))
        # the slabs as the kernel cuts them: workgroup w takes rows 4 w + wave + 4 nparts k
        nparts = self.rownorm_parts(rows, C)
        p = part.reshape(nparts, 2, C)
        p.zero_()
        owner = (torch.arange(rows) // 4) % nparts
        p[:, 0].index_add_(0, owner.to(p.device), g * xh)
        p[:, 1].index_add_(0, owner.to(p.device), g)

    def relu_drop_fwd(self, h, a, n, p_drop=0.0, seed=0):
        keep, kinv = self._rownorm_keep(n, p_drop, seed)
        v = torch.clamp(h.reshape(n), min=0)
        a.reshape(n).copy_(v if keep is None else torch.where(keep.to(v.device), v * kinv, torch.zeros_like(v)))

    def relu_drop_bwd(self, dy, a, dh, n, p_drop=0.0):
        g = dy.reshape(n)
        dh.reshape(n).copy_(torch.where(a.reshape(n) != 0, g * (1.0 / (1.0 - p_drop)), torch.zeros_like(g)))

    # ------------------------------------------------------------------ stand-alone gLN
    def gln_stats(self, x, stats, B, C, T, ldt):
        v = x.reshape(B, C, ldt)[:, :, :T]
        _acc(stats, v.sum((1, 2)), (v * v).sum((1, 2)))

    def gln_apply(self, x, stats, gamma, beta, y, B, C, T, ldt, count, eps):
        mu, rstd = _mu_rstd(stats, count, eps, x.dtype)
        sc = gamma.view(1, C, 1) * rstd
        v = x.reshape(B, C, ldt) * sc + (beta.view(1, C, 1) - mu * sc)
        v[:, :, T:] = 0
        y.reshape(B, C, ldt).copy_(v)

    def gln_bwd_rowsums(self, dy, x, rowpart, B, C, T, ldt):
        ntile = (ldt + 1023) // 1024
        rp = rowpart.reshape(B, C, ntile, 2)
        rp.zero_()
        g = dy.reshape(B, C, ldt)[:, :, :T]
        rp[:, :, 0, 0] = g.sum(2)
        rp[:, :, 0, 1] = (g * x.reshape(B, C, ldt)[:, :, :T]).sum(2)

    def gln_bwd_apply(self, dy, x, stats, gamma, bsum, dx, B, C, T, ldt, count, eps):
        mu, rstd = _mu_rstd(stats, count, eps, x.dtype)
        xh = (x.reshape(B, C, ldt) - mu) * rstd
        v = rstd * (gamma.view(1, C, 1) * dy.reshape(B, C, ldt) - bsum[:, 0].view(B, 1, 1) - xh * bsum[:, 1].view(B, 1, 1))
        v[:, :, T:] = 0
        dx.reshape(B, C, ldt).copy_(v)

    def repack(self, src, ld_src, dst, ld_dst, rows, T):
        d = dst.reshape(rows, ld_dst)
        d.zero_()
        d[:, :T] = src.reshape(rows, ld_src)[:, :T]

    @staticmethod
    def _depthwise(x3, w, bias, C, Tin, Tout, Kw, stride, pad, dil):
        """y[to] = bias + sum_k w[k] xpad[to*stride + k*dil - pad] for to < Tout, x zero outside [0, Tin) -- the kernel's contract: ANY Tout
        (the TCN layers ask for Tout = Tin with pad = (Kw-1) dil: all of the padding on the left)"""
        need = (Tout - 1) * stride + (Kw - 1) * dil + 1              # padded input frames the Tout outputs reach
        right = max(0, need - pad - Tin)
        xp = torch.nn.functional.pad(x3, (pad, right))
        r = torch.nn.functional.conv1d(xp, w.reshape(C, 1, Kw), None if bias is None else bias.reshape(C), stride=stride, dilation=dil, groups=C)
        return r[:, :, :Tout]

    def depthwise_fwd(self, x, w, bias, y, B, C, Tin, Tout, Kw, stride, pad, dil):
        y.reshape(B, C, Tout).copy_(self._depthwise(x.reshape(B, C, Tin), w, bias, C, Tin, Tout, Kw, stride, pad, dil))

    def depthwise_bwd_input(self, dy, w, dx, B, C, Tin, Tout, Kw, stride, pad, dil):
        with torch.enable_grad():
            x0 = torch.zeros(B, C, Tin, dtype=dy.dtype, requires_grad=True)
            r = self._depthwise(x0, w.detach(), None, C, Tin, Tout, Kw, stride, pad, dil)
            gx = torch.autograd.grad(r, x0, dy.detach().reshape(B, C, Tout))[0]
        dx.reshape(B, C, Tin).copy_(gx)

    def depthwise_bwd_weight(self, dy, x, partial, B, C, Tin, Tout, Kw, stride, pad, dil):
        g = dy.reshape(B, C, Tout)
        xp = torch.zeros(B, C, Tin + 2 * pad + Kw * dil + Tout * stride, dtype=x.dtype)
        xp[:, :, pad:pad + Tin] = x.reshape(B, C, Tin)
        out = partial.reshape(B, C, Kw + 1)
        to = torch.arange(Tout)
        for k in range(Kw):
            out[:, :, k] = (g * xp[:, :, to * stride + k * dil]).sum(2)
        out[:, :, Kw] = g.sum(2)

    def segment(self, x, out, rows, T, ldt, S, chunk, hop, pad_left):
        total = (S - 1) * hop + chunk
        xp = torch.zeros(rows, total + pad_left + T, dtype=x.dtype)
        xp[:, pad_left:pad_left + T] = x.reshape(rows, ldt)[:, :T]
        idx = (torch.arange(S) * hop).view(S, 1) + torch.arange(chunk).view(1, chunk)
        out.reshape(rows, S, chunk).copy_(xp[:, idx])

    def overlap_add(self, y, out, rows, T, ldt, S, chunk, hop, pad_left):
        total = (S - 1) * hop + chunk
        acc = torch.zeros(rows, total + T, dtype=y.dtype)
        idx = ((torch.arange(S) * hop).view(S, 1) + torch.arange(chunk).view(1, chunk)).reshape(-1)
        acc.index_add_(1, idx, y.reshape(rows, S * chunk))
        o = out.reshape(rows, ldt)
        o.zero_()
        o[:, :T] = acc[:, pad_left:pad_left + T]

    # ------------------------------------------------------------------ losses
    def sisdr_dots(self, est, tgt, dots, tt, xx, B, n, T, all_pairs):
        e, t = est.reshape(B, n, T).double(), tgt.reshape(B, n, T).double()
        full = torch.einsum("bit,bjt->bij", e, t)
        if not all_pairs:
            full = torch.diag_embed(torch.diagonal(full, dim1=1, dim2=2))
        dots.reshape(B, n, n).add_(full)
        tt.reshape(B, n).add_((t * t).sum(2))
        xx.reshape(B, n).add_((e * e).sum(2))

    @staticmethod
    def _terms(a, ttv, xxv, eps):
        c = ttv + eps
        alpha = a / c
        S = alpha * alpha * ttv + eps
        Nn = (alpha * alpha * ttv - 2 * alpha * a + xxv).clamp_min(0) + eps
        return alpha, c, S, Nn

    def sisdr_from_dots(self, dots, tt, xx, out, B, n, all_pairs, eps):
        a = dots.reshape(B, n, n)
        alpha, c, S, Nn = self._terms(a, tt.reshape(B, 1, n), xx.reshape(B, n, 1), eps)
        v = 10.0 * torch.log10(S / Nn)
        if not all_pairs:
            v = torch.diag_embed(torch.diagonal(v, dim1=1, dim2=2))
        out.reshape(B, n, n).copy_(v.to(out.dtype))

    def sisdr_bwd(self, est, tgt, dots, tt, xx, gw, d_est, B, n, T, all_pairs, eps):
        a = dots.reshape(B, n, n)
        ttv, xxv = tt.reshape(B, 1, n), xx.reshape(B, n, 1)
        alpha, c, S, Nn = self._terms(a, ttv, xxv, eps)
        Kc = 10.0 / math.log(10.0)
        ct = Kc * (2 * alpha * ttv / (c * S) - ((2 * alpha * ttv - 2 * a) / c - 2 * alpha) / Nn)
        ce = Kc * (-2.0 / Nn)
        g = gw.reshape(B, n, n).double()
        if not all_pairs:
            g = torch.diag_embed(torch.diagonal(g, dim1=1, dim2=2))
        cT = g * ct
        cE = (g * ce).sum(2)
        out = torch.einsum("bij,bjt->bit", cT, tgt.reshape(B, n, T).double()) + cE.view(B, n, 1) * est.reshape(B, n, T).double()
        d_est.reshape(B, n, T).copy_(out.to(d_est.dtype))

    def pit_search(self, val, perms, P, n, B, maximize, use_mean, best_val, best_idx):
        v = val.reshape(B, n, n)
        sc = torch.stack([sum(v[:, k, int(perms[p, k])] for k in range(n)) for p in range(P)], 1)
        if use_mean:
            sc = sc / n
        bv, bi = (sc.max(1) if maximize else sc.min(1))
        best_val.copy_(bv)
        best_idx.copy_(bi)

    def sinkhorn_fwd(self, C, zwork, loss, P, B, n, coldness, iters):
        Cd = C.reshape(B, n, n).double()
        Z = -coldness * Cd
        zw = zwork.reshape(B, 2 * iters + 1, n, n)
        zw[:, 0] = Z
        for h in range(1, 2 * iters + 1):
            Z = Z - torch.logsumexp(Z, dim=1 if (h & 1) else 2, keepdim=True)
            zw[:, h] = Z
        Pm = torch.exp(Z)
        P.reshape(B, n, n).copy_(Pm.to(P.dtype))
        loss.copy_(((Cd + Z / coldness) * Pm).sum((1, 2)).to(loss.dtype))

    def sinkhorn_bwd(self, C, zwork, dloss, dC, B, n, coldness, iters):
        Cd = C.reshape(B, n, n).double()
        zw = zwork.reshape(B, 2 * iters + 1, n, n)
        Zf = zw[:, 2 * iters]
        g = dloss.reshape(B, 1, 1).double()
        Pm = torch.exp(Zf)
        dZ = g * Pm * (1.0 / coldness + Cd + Zf / coldness)
        for h in range(2 * iters, 0, -1):
            dim = 1 if (h & 1) else 2
            dZ = dZ - torch.exp(zw[:, h]) * dZ.sum(dim, keepdim=True)
        dC.reshape(B, n, n).copy_((g * Pm - coldness * dZ).to(dC.dtype))

    # ------------------------------------------------------------------ distance criteria
    def rowdiff_sums(self, x, t, sums, rows, T):
        d = (x.reshape(rows, T) - t.reshape(rows, T)).double()
        sums.reshape(rows, 3).copy_(torch.stack([d.abs().sum(1), (d * d).sum(1), (t.reshape(rows, T).double() ** 2).sum(1)], dim=1))

    def rowdiff_bwd(self, x, t, c_abs, c_sq, dx, rows, T):
        d = x.reshape(rows, T) - t.reshape(rows, T)
        g = torch.zeros_like(d)
        if c_abs is not None:
            g = g + c_abs.reshape(rows, 1) * torch.sign(d)
        if c_sq is not None:
            g = g + c_sq.reshape(rows, 1) * d
        dx.reshape(rows, T).copy_(g)

    # ------------------------------------------------------------------ optimiser
    def sqnorm(self, g, out, n):
        out += (g.reshape(-1)[:n].double() ** 2).sum()

    def lstm_fwd(self, xg, w_hh, h_out, gates, cstate, nseq, L, H, reverse):
        interleaved = bool(int(reverse) & 0x400)
        reverse = int(reverse) & 0xff        # bits 8-9 choose between the device's two sweep kernels: same arithmetic
        if interleaved:                # h_out is ONE (nseq, L, 2H) buffer: forward direction in columns [0, H), reversed in [H, 2H)
            slabs = torch.empty(2, nseq, L, H, dtype=h_out.dtype)
            self.lstm_fwd(xg, w_hh, slabs, gates, cstate, nseq, L, H, 2)
            h_out.reshape(nseq, L, 2 * H)[:] = torch.cat([slabs[0], slabs[1]], dim=2)
            return
        if int(reverse) == 2:          # both directions: two slabs per buffer
            for d in range(2):
                self.lstm_fwd(xg.reshape(2, -1)[d], w_hh.reshape(2, 4 * H, H)[d], h_out.reshape(2, -1)[d],
                              None if gates is None else gates.reshape(2, -1)[d], None if cstate is None else cstate.reshape(2, -1)[d],
                              nseq, L, H, d)
            return
        xg3 = xg.reshape(nseq, L, 4 * H)
        h = torch.zeros(nseq, H, dtype=xg.dtype)
        c = torch.zeros(nseq, H, dtype=xg.dtype)
        ho = h_out.reshape(nseq, L, H)
        for step in range(L):
            t = L - 1 - step if reverse else step
            a = xg3[:, t] + h @ w_hh.t()
            i, f, g, o = torch.sigmoid(a[:, :H]), torch.sigmoid(a[:, H:2 * H]), torch.tanh(a[:, 2 * H:3 * H]), torch.sigmoid(a[:, 3 * H:])
            c = f * c + i * g
            h = o * torch.tanh(c)
            ho[:, t] = h
            if gates is not None:
                gates.reshape(nseq, L, 4 * H)[:, t] = torch.cat([i, f, g, o], dim=1)
            if cstate is not None:
                cstate.reshape(nseq, L, H)[:, t] = c

    def lstm_bwd(self, dh_out, gates, cstate, w_hh, dxg, nseq, L, H, reverse):
        interleaved = bool(int(reverse) & 0x400)
        reverse = int(reverse) & 0xff
        if interleaved:                # dh_out is ONE (nseq, L, 2H) buffer
            d3 = dh_out.reshape(nseq, L, 2 * H)
            self.lstm_bwd(torch.stack([d3[..., :H], d3[..., H:]]).contiguous(), gates, cstate, w_hh, dxg, nseq, L, H, 2)
            return
        if int(reverse) == 2:
            for d in range(2):
                self.lstm_bwd(dh_out.reshape(2, -1)[d], gates.reshape(2, -1)[d], cstate.reshape(2, -1)[d], w_hh.reshape(2, 4 * H, H)[d],
                              dxg.reshape(2, -1)[d], nseq, L, H, d)
            return
        G = gates.reshape(nseq, L, 4 * H)
        C = cstate.reshape(nseq, L, H)
        dho = dh_out.reshape(nseq, L, H)
        out = dxg.reshape(nseq, L, 4 * H)
        dhr = torch.zeros(nseq, H, dtype=G.dtype)
        dc = torch.zeros(nseq, H, dtype=G.dtype)
        for step in range(L):
            t = step if reverse else L - 1 - step
            tp = t + 1 if reverse else t - 1
            cp = C[:, tp] if 0 <= tp < L else torch.zeros(nseq, H, dtype=G.dtype)
            i, f, g, o = G[:, t, :H], G[:, t, H:2 * H], G[:, t, 2 * H:3 * H], G[:, t, 3 * H:]
            dh = dho[:, t] + dhr
            tc = torch.tanh(C[:, t])
            dcc = dh * o * (1 - tc * tc) + dc
            da = torch.cat([dcc * g * i * (1 - i), dcc * cp * f * (1 - f), dcc * i * (1 - g * g), dh * tc * o * (1 - o)], dim=1)
            out[:, t] = da
            dc = dcc * f
            dhr = da @ w_hh

    def chunk_to_tokens(self, x, y, B, F, S, K, inter):
        v = x.reshape(B, F, S, K)
        y.reshape(-1)[:] = (v.permute(0, 3, 2, 1) if inter else v.permute(0, 2, 3, 1)).reshape(-1)

    def tokens_to_chunk(self, y, x, B, F, S, K, inter):
        v = y.reshape(B, K, S, F).permute(0, 3, 2, 1) if inter else y.reshape(B, S, K, F).permute(0, 3, 1, 2)
        x.reshape(-1)[:] = v.reshape(-1)

    # ---- token-major dense layers (csrc/linear.hip)
    def linear_fwd(self, x, w, bias, bias2, y, ntok, K, N):
        out = x.reshape(ntok, K) @ w.reshape(N, K).t()
        if bias is not None:
            out = out + bias
        if bias2 is not None:
            out = out + bias2
        y.reshape(ntok, N)[:] = out

    def linear_bwd_input(self, dy, w, dx, ntok, K, N, accumulate):
        out = dy.reshape(ntok, N) @ w.reshape(N, K)
        if accumulate:
            dx.reshape(ntok, K)[:] += out
        else:
            dx.reshape(ntok, K)[:] = out

    def linear_bwd_weight(self, dy, x, ldx, partial, partial_bias, ntok, K, N, L, shift, nslab):
        assert x.dim() == 2 and x.stride(0) == ldx and x.stride(1) == 1
        d2, x2 = dy.reshape(ntok, N), x.reshape(ntok, K)
        if shift:
            xs = torch.zeros_like(x2).reshape(ntok // L, L, K)
            xv = x2.reshape(ntok // L, L, K)
            if shift < 0:
                xs[:, 1:] = xv[:, :-1]
            else:
                xs[:, :-1] = xv[:, 1:]
            x2 = xs.reshape(ntok, K)
        chunks = (ntok + 31) // 32
        per = (chunks + nslab - 1) // nslab * 32
        for s_ in range(nslab):
            lo, hi = min(ntok, s_ * per), min(ntok, (s_ + 1) * per)
            partial.reshape(nslab, N, K)[s_] = d2[lo:hi].t() @ x2[lo:hi]
            if partial_bias is not None:
                partial_bias.reshape(nslab, N)[s_] = d2[lo:hi].sum(0)

    def adam_step_dev(self, p, g, m, v, sqnorm, n, lr_dev, step_dev, beta1, beta2, eps, weight_decay, max_norm, grad_scale):
        step_dev += 1
        self.adam_step(p, g, m, v, sqnorm, n, float(lr_dev[0]), beta1, beta2, eps, weight_decay, max_norm, grad_scale, int(step_dev[0]))

    def adam_step(self, p, g, m, v, sqnorm, n, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, step):
        coef = grad_scale
        if max_norm > 0:
            total = grad_scale * float(torch.sqrt(sqnorm.reshape(-1)[0]))
            coef *= min(1.0, max_norm / (total + 1e-6))
        g.mul_(coef)
        gi = g + weight_decay * p if weight_decay != 0 else g
        m.mul_(beta1).add_(gi, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2s = math.sqrt(1 - beta2 ** step)
        p.sub_((lr / bc1) * (m / (v.sqrt() / bc2s + eps)))
