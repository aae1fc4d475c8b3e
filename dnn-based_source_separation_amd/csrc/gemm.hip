// Pointwise (1x1) convolutions of the Conv-TasNet separator as fp32 MFMA GEMMs for gfx950.
//
//   forward / input-gradient :  Y[b][m][t] = epi( sum_k A[m][k] * pro(X[b][k][t]) + bias[m] )     (sep_pw_gemm)
//   weight gradient          :  dW[m][n]   = sum_{b,t} G[b][m][t] * pro(X[b][n][t])               (sep_pw_wgrad)
//
// Replaces nn.Conv1d(kernel_size=1) of reference src/models/tdcn.py:86,173,175 and
// src/models/conv_tasnet.py:335,341 together with the elementwise ops the reference runs as separate
// ATen kernels around them (PReLU, gLN apply, residual/skip adds, sigmoid, their backward forms).
//
// Layout of this file:
//   gemm_epilogue<EF>             LDS-transposed float4 epilogue shared by both GEMM kernels (bias, PReLU statistics,
//                                 residual / skip accumulate, sigmoid, PReLU backward, gLN-backward row sums)
//   pw_gemm_kernel                register-staged GEMM; fallback for contractions that are not a multiple of 16
//   pw_gemm_direct_kernel<...,AR> THE fast path, LDS-DMA ring, prologue (PReLU / gLN / gLN backward) on the B fragments.
//                                 AR = 2 (SEP_ARITH_F16X3, default): fp32 products by a scaled two-part fp16 split on
//                                 v_mfma_f32_32x32x16_f16 (one scale for A, one per column of X kept per lane);
//                                 AR = 1 (SEP_ARITH_BF16X6): fp32 products by exact three-way bf16 split on
//                                 v_mfma_f32_32x32x16_bf16; both: 3-stage ring, 3 workgroups per CU, whole-chunk steps;
//                                 AR = 0 (SEP_ARITH_F32): v_mfma_f32_32x32x2_f32, 2-stage ring, 4 workgroups per CU,
//                                 half-chunk software pipeline -- design notes at the kernel
//   pw_wgrad_kernel               register-staged weight gradient; fallback (g_mul = decoder basis gradient)
//   pw_wgrad_split_kernel<X>      weight gradient in the split arithmetic: 4-wave workgroups, 3 per CU, 3-stage ring
//   pw_wgrad_direct_kernel<X>     weight gradient on the fp32 MFMA: 8-wave workgroups, two wave groups split the contraction
//   reduce_slabs / f64_to_f32     deterministic second stages
// Common to all: 128x128 output tile, each wave a 64x64 sub-tile = 2x2 32x32 MFMA accumulators (64 VGPRs);
// the normalised tensors v1, v2 of the reference never exist in HBM; workgroups that share an X column tile are placed
// on the same XCD (blockIdx % 8) so the tile is fetched from HBM once and re-read from that L2.
#include "gemm_common.hpp"
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int BK = 32;       // contraction rows per chunk (forward / dgrad): 64 MFMAs per wave between barriers
constexpr int LDA_S = 132;   // allocated As[k][m] row stride (floats); 16B aligned rows for the float4 (transposed-A) writes
constexpr int LDA_NT = 129;  // row stride actually used by the non-transposed path: odd -> conflict-free transposing scalar writes
constexpr int LDB_S = 128;

struct __attribute__((aligned(16))) GemmSmem {
    float As[2][BK * LDA_S];
    float Bs[2][BK * LDB_S];
    double red[8];
};

// One 32-deep chunk of the contraction for a wave's 64x64 tile: 16 k-steps x 4 v_mfma_f32_32x32x2_f32.
// Operand fragments are fetched from LDS in groups of 4 k-steps, one group ahead of the MFMAs that consume
// them (two static register sets), so the LDS latency is paid once per chunk instead of once per k-step and
// the matrix pipe sees 64 back-to-back MFMAs.  A/B are k-major: element (k, i) at base[k*ld + i].
__device__ __forceinline__ void mfma_chunk32(const float* __restrict__ Ab, const int lda, const float* __restrict__ Bb,
                                             const int ldb, const int aoff, const int boff, const int lk,
                                             f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11) {
    float fa0[4][2], fb0[4][2], fa1[4][2], fb1[4][2];
#define SEP_LOAD_FRAGS(FA, FB, g)                                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                    \
        const int ka = 2 * (4 * (g) + ks) + lk;                           \
        FA[ks][0] = Ab[ka * lda + aoff];                                  \
        FA[ks][1] = Ab[ka * lda + aoff + 32];                             \
        FB[ks][0] = Bb[ka * ldb + boff];                                  \
        FB[ks][1] = Bb[ka * ldb + boff + 32];                             \
    }
#define SEP_MFMA_GROUP(FA, FB)                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                    \
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[ks][0], FB[ks][0], c00, 0, 0, 0);   \
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[ks][0], FB[ks][1], c01, 0, 0, 0);   \
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[ks][1], FB[ks][0], c10, 0, 0, 0);   \
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[ks][1], FB[ks][1], c11, 0, 0, 0);   \
    }
    SEP_LOAD_FRAGS(fa0, fb0, 0)
    SEP_LOAD_FRAGS(fa1, fb1, 1)
    SEP_MFMA_GROUP(fa0, fb0)
    SEP_LOAD_FRAGS(fa0, fb0, 2)
    SEP_MFMA_GROUP(fa1, fb1)
    SEP_LOAD_FRAGS(fa1, fb1, 3)
    SEP_MFMA_GROUP(fa0, fb0)
    SEP_MFMA_GROUP(fa1, fb1)
#undef SEP_LOAD_FRAGS
#undef SEP_MFMA_GROUP
}



__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(const sep_gemm_desc d) {
    __shared__ GemmSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    const int NR = (d.M + BM - 1) / BM;
    const int ntile_t = d.ldt / BN;
    const int NC = d.B * ntile_t;
    // XCD-aware decode: all row tiles of one column tile land on the same XCD (blockIdx % 8)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int rt = j % NR;
    const int ct = (j / NR) * 8 + xcd;
    if (ct >= NC) return;
    const int b = ct / ntile_t;
    const int t0 = (ct % ntile_t) * BN;
    const int m0 = rt * BM;
    if (t0 >= d.T && d.pro_mode != SEP_PRO_GLN_BWD) {
        // whole tile lies in the pad region: outputs are zero there (never read-modify-written)
        for (int i = tid; i < BM * (BN / 4); i += 256) {
            const int r = i / (BN / 4), c4 = i % (BN / 4);
            const int row = m0 + r;
            if (row >= d.M) continue;
            float* dst;
            size_t idx;
            if (d.m_split && row >= d.m_split) {
                dst = d.Y2; idx = ((size_t)b * (d.M - d.m_split) + (row - d.m_split)) * d.ldt;
            } else {
                dst = d.Y; idx = ((size_t)b * (d.m_split ? d.m_split : d.M) + row) * d.ldt;
            }
            st4(dst + idx + t0 + 4 * c4, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        if (d.epi_flags & SEP_EPI_ROWSUMS) {
            for (int i = tid; i < BM * 2; i += 256) {
                const int row = m0 + (i >> 1);
                if (row < d.M) {
                    float* rp = d.epi_rowpart + (((size_t)b * d.M + row) * (d.ldt / 64) + (t0 / 64) + (i & 1)) * 2;
                    rp[0] = 0.f; rp[1] = 0.f;
                }
            }
        }
        return;
    }

    // ---- per-block prologue constants ------------------------------------------------
    float mu = 0.f, rstd = 1.f, alpha_p = 0.f, mg = 0.f, mgx = 0.f;
    const int pro = d.pro_mode;
    if (pro == SEP_PRO_GLN || pro == SEP_PRO_GLN_PRELU || pro == SEP_PRO_GLN_BWD) gln_mu_rstd(d.pro_stats + (size_t)b * SEP_STATS_SLOTS * 2, d.count, d.eps, mu, rstd);
    if (pro == SEP_PRO_PRELU || pro == SEP_PRO_GLN_PRELU || pro == SEP_PRO_GLN_BWD) alpha_p = d.pro_alpha[0];
    if (pro == SEP_PRO_GLN_BWD) {
        if (d.pro_bacc) gln_bwd_means(d.pro_bacc + (size_t)b * SEP_STATS_SLOTS * 2, d.pro_stats + (size_t)b * SEP_STATS_SLOTS * 2, d.count, d.eps, mg, mgx);
        else { mg = d.pro_bsum[2 * b]; mgx = d.pro_bsum[2 * b + 1]; }
    }
    float dalpha_pro = 0.f;

    // ---- thread -> staging coordinates -------------------------------------------------
    const int br = tid >> 5, bc4 = tid & 31;   // B tile: rows br + 8i (i<4) ; float4 column bc4
    const int am = tid >> 3, ak4 = tid & 7;    // A tile (non-trans): rows am + 32i (i<4) ; float4 along k
    const int nk = d.K / BK;
    const int lda_s = d.trans_a ? LDA_S : LDA_NT;

    float4 ra[4], rb[4], rx[4];
    float pgam[4], pbet[4];     // gLN affine of the rows being staged, fetched together with the tile (one chunk ahead)

    auto load_global = [&](int kc) {
        const int k0 = kc * BK;
        const float* Xs = d.X;
        const float* As_ = d.A;
        int krow = k0, Ksrc = d.K;
        bool second = false;
        if (d.k_split) {
            if (k0 >= d.k_split) { Xs = d.X2; As_ = d.A2; krow = k0 - d.k_split; Ksrc = d.K - d.k_split; second = true; }
            else { Ksrc = d.k_split; }
        }
        (void)second;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t off = ((size_t)b * Ksrc + krow + br + 8 * i) * d.ldt + t0 + 4 * bc4;
            rb[i] = ld4(Xs + off);
            if (pro == SEP_PRO_GLN_BWD) rx[i] = ld4(d.pro_aux + off);
            if (pro >= SEP_PRO_GLN) {
                pgam[i] = d.pro_gamma[k0 + br + 8 * i];
                pbet[i] = (pro == SEP_PRO_GLN_BWD) ? 0.f : d.pro_beta[k0 + br + 8 * i];
            }
        }
        if (d.trans_a) {
            // A is [K][M]: row k, float4 along m
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mm = m0 + 4 * bc4;
                if (mm < d.M) ra[i] = ld4(As_ + (size_t)(krow + br + 8 * i) * d.M + mm);
                else ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mm = m0 + am + 32 * i;
                if (mm < d.M) ra[i] = ld4(As_ + (size_t)mm * Ksrc + krow + 4 * ak4);
                else ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    auto store_lds = [&](int kc, int buf) {
        const int k0 = kc * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
            const int kg = k0 + br + 8 * i;   // global contraction row (parameter index of gamma/beta)
            if (pro == SEP_PRO_PRELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = prelu_f(v[q], alpha_p);
            } else if (pro == SEP_PRO_GLN || pro == SEP_PRO_GLN_PRELU) {
                const float sc = pgam[i] * rstd;
                const float sh = pbet[i] - mu * sc;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = (pro == SEP_PRO_GLN_PRELU) ? prelu_f(v[q], alpha_p) : v[q];
                    v[q] = u * sc + sh;
                }
            } else if (pro == SEP_PRO_GLN_BWD) {
                const float a4[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
                const float gk = pgam[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = t0 + 4 * bc4 + q;
                    const float a = a4[q];
                    const float u = prelu_f(a, alpha_p);
                    const float xh = (u - mu) * rstd;
                    const float du = rstd * (gk * v[q] - mg - xh * mgx);
                    float da = du * prelu_grad(a, alpha_p);
                    if (t >= d.T) da = 0.f;
                    else if (a <= 0.f && rt == 0) dalpha_pro += du * a;
                    v[q] = da;
                }
                if (rt == 0) {
                    const size_t off = ((size_t)b * d.K + kg) * d.ldt + t0 + 4 * bc4;
                    st4(d.pro_store + off, make_float4(v[0], v[1], v[2], v[3]));
                }
            }
            st4(&sm.Bs[buf][(br + 8 * i) * LDB_S + 4 * bc4], make_float4(v[0], v[1], v[2], v[3]));
        }
        if (d.trans_a) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st4(&sm.As[buf][(br + 8 * i) * LDA_S + 4 * bc4], ra[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mm = am + 32 * i;
                sm.As[buf][(4 * ak4 + 0) * LDA_NT + mm] = ra[i].x;
                sm.As[buf][(4 * ak4 + 1) * LDA_NT + mm] = ra[i].y;
                sm.As[buf][(4 * ak4 + 2) * LDA_NT + mm] = ra[i].z;
                sm.As[buf][(4 * ak4 + 3) * LDA_NT + mm] = ra[i].w;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_global(0);
    store_lds(0, 0);
    __syncthreads();

    const int lk = lane >> 5, l31 = lane & 31;
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nk) load_global(kc + 1);
        mfma_chunk32(sm.As[cur], lda_s, sm.Bs[cur], LDB_S, wr * 64 + l31, wc * 64 + l31, lk, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        if (kc + 1 < nk) store_lds(kc + 1, cur ^ 1);
        __syncthreads();
    }

    __syncthreads();                                    // K-loop LDS reads are done: the staging area becomes the transpose buffer
    #ifdef SEP_PROF
    const int prof_slot = -1;
#endif
    gemm_epilogue<-1>(d, acc, b, m0, t0, wr, wc, lk, l31, tid, &sm.As[0][0], sm.red, BN, true PROF_PASS);
    if (pro == SEP_PRO_GLN_BWD && rt == 0) {
        const double s = block_sum_256<double>((double)dalpha_pro, sm.red);
        if (tid == 0) atomicAdd(d.pro_dalpha, s);
    }
}



// ======================================================================================
// The fast GEMM: direct-to-LDS, 2-stage ring, high occupancy.
//
// Why this shape (all measured on MI355X, tools/gemm_bench.py + rocprofv3):
//  * a register-staged kernel (above, kept for odd shapes) is latency-bound: removing all its MFMAs shortened it by < 35 %;
//  * a 4-stage DMA ring with counted vmcnt, fragment double-buffering and a hand-ordered instruction stream reached
//    120 TF/s on long contractions but only 65 TF/s at K = 128, and 1 -> 2 workgroups per CU was worth +25...45 %:
//    a wave issues in order, so while it runs its DMA issue / ds_reads / prologue / epilogue the matrix pipe needs
//    OTHER waves.  Trading ring depth and register double-buffering for occupancy (2 stages = 32 KiB LDS, <= 128
//    VGPRs => FOUR workgroups per CU) beat that kernel on every shape of the model, with far simpler code;
//  * with one chunk of lead (a chunk's DMA is issued one MFMA burst x 4 co-resident waves before it is needed) the
//    waits are plain vmcnt(0), so other memory traffic in flight (epilogue stores, the gLN-backward store-back) is harmless.
//
//   A k-major ([K][M], input-gradient form): rows DMA'd verbatim, fragments by conflict-free ds_read_b32.
//   A row-major ([M][K], forward form): 16-byte granules DMA'd with an XOR swizzle on the SOURCE address (the LDS image
//   of a DMA is lane-linear), fragments by two ds_read_b128 per 32-row block; the contraction index is permuted (lane
//   half lk owns k = 8*lk .. 8*lk+7 of the chunk) identically for A and B, which a sum allows.
//   The elementwise prologue acts on the B fragments after the ds_read; its per-row affine sits in LDS.
//   PRO == GLN_BWD streams a third operand (the pre-activation a), forms d(pre-activation) on the fragments, and the
//   wr == 0 waves of row tile 0 write it back for the weight-gradient GEMM and accumulate the PReLU slope gradient.
// ======================================================================================
constexpr int OMAXK = 512;      // rows of the per-row affine table in LDS (3 stages x 3 workgroups per CU must fit 160 KiB)

template <bool AUX, int NS = 2>
struct __attribute__((aligned(16))) DirectSmem {
    float As[NS][DK * 128];
    float Bs[NS][DK * 128];
    float Cs[AUX ? 2 : 1][AUX ? DK * 128 : 4];
    float sc[OMAXK];
    float sh[AUX ? 4 : OMAXK];
    double red[8];
};

#ifndef SEP_GLN_OCC
#define SEP_GLN_OCC 4       // measured: 4 blocks/CU with ~40 B of spill (outside the hot loop) = 3 blocks/CU spill-free (182 vs 186 us on the heads GEMM)
#endif
// AR selects the arithmetic of the contraction:
//   0  v_mfma_f32_32x32x2_f32 on the fp32 fragments (157 TF/s peak).
//   1  fp32 by exact three-way bf16 splitting: every fragment value x = hi + mid + lo with each part a truncated bf16
//      (8 + 8 + 8 significand bits: the sum is EXACT), and x*y = hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi
//      (+ three terms <= 2^-24 |xy| that are dropped) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 MFMAs of
//      32 cycles per 16-deep chunk and tile pair instead of 8 of 64.  Error per product <= 2^-23 relative, i.e. the
//      rounding of an fp32 multiply; measured against fp64 it is no worse than the fp32 MFMA path (tests).  The
//      lane half lk already owns k = 8*lk .. 8*lk+7 of a chunk, which is the operand layout of the bf16 instruction.

// two fp32 values -> their (hi, mid, lo) bf16 parts packed {x0 low half, x1 high half}: 11 VALU instructions
__device__ __forceinline__ void split3_pair(const float x0, const float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}
// the 8 chunk values of one 32-row (32-column) block -> three packed operands
__device__ __forceinline__ void split3_frag(const float (&lo4)[4], const float (&hi4)[4], u32x4_t (&out)[3]) {
    unsigned p[3][4];
    split3_pair(lo4[0], lo4[1], p[0][0], p[1][0], p[2][0]);
    split3_pair(lo4[2], lo4[3], p[0][1], p[1][1], p[2][1]);
    split3_pair(hi4[0], hi4[1], p[0][2], p[1][2], p[2][2]);
    split3_pair(hi4[2], hi4[3], p[0][3], p[1][3], p[2][3]);
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = u32x4_t{p[q][0], p[q][1], p[q][2], p[q][3]};
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4_t a, const u32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// acc += A * B on the split operands (6 of the 9 part products)
__device__ __forceinline__ void mfma_split6(const u32x4_t (&a)[3], const u32x4_t (&b)[3], f32x16& acc) {
    acc = mfma_bf16(a[0], b[2], acc);
    acc = mfma_bf16(a[2], b[0], acc);
    acc = mfma_bf16(a[1], b[1], acc);
    acc = mfma_bf16(a[0], b[1], acc);
    acc = mfma_bf16(a[1], b[0], acc);
    acc = mfma_bf16(a[0], b[0], acc);
}


template <bool TRANS_A, int PRO, bool SPLIT, int EF, int AR = 0>
__global__ __launch_bounds__(256, (AR >= 1 ? 3 : PRO == SEP_PRO_GLN_BWD ? 3 : PRO >= SEP_PRO_GLN ? SEP_GLN_OCC : 4)) void pw_gemm_direct_kernel(const sep_gemm_desc d) {
    constexpr bool P_PRELU = PRO == SEP_PRO_PRELU || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_GLN = PRO == SEP_PRO_GLN || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_BWD = PRO == SEP_PRO_GLN_BWD;
    static_assert(AR == 0 || PRO != SEP_PRO_GLN_BWD, "the split arithmetic has no GLN_BWD form");
    constexpr int NS = AR >= 1 ? 3 : 2;          // ring depth: the split paths consume a chunk ~3x faster, so they prefetch two ahead          // ring depth: the bf16 path consumes a chunk ~3x faster, so it prefetches two ahead
    __shared__ DirectSmem<P_BWD, NS> sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the DMA's LDS base goes to M0 without a waterfall loop
    const int wr = wid >> 1, wc = wid & 1;
    const int lk = lane >> 5, l31 = lane & 31;

    const int NR = (d.M + BM - 1) / BM;
    const int ntile_t = d.ldt / BN;
    const int NC = d.B * ntile_t;
    // XCD-aware decode: all row tiles of one column tile land on the same XCD (blockIdx % 8) and share X through its L2
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int rt = j % NR;
    const int ct = (j / NR) * 8 + xcd;
    if (ct >= NC) return;
    const int b = ct / ntile_t;
    const int t0 = (ct % ntile_t) * BN;
    const int m0 = rt * BM;
    const int nk = d.K / DK;
#ifdef SEP_PROF
    const int prof_slot = bid == 8 ? 0 : bid == 1500 ? 1 : bid == 1501 ? 2 : bid == (int)gridDim.x - 9 ? 3 : -1;
#endif
    PROF_STAMP(0);
#ifdef SEP_PROF
    if (tid == 0 && bid < 8192) g_blk_start[bid] = wall_clock64();
#endif

    // per-row affine of the prologue, once per workgroup
    float alpha_p = 0.f, mu = 0.f, rstd = 1.f, mg = 0.f, mgx = 0.f;
    if (P_PRELU || P_BWD) alpha_p = d.pro_alpha[0];
    if (P_GLN || P_BWD) {
        gln_mu_rstd(d.pro_stats + (size_t)b * SEP_STATS_SLOTS * 2, d.count, d.eps, mu, rstd);
        for (int k = tid; k < d.K; k += 256) {
            if (P_BWD) sm.sc[k] = d.pro_gamma[k];
            else {
                const float scv = d.pro_gamma[k] * rstd;
                sm.sc[k] = scv;
                sm.sh[k] = d.pro_beta[k] - mu * scv;
            }
        }
    }
    if (P_BWD) {
        if (d.pro_bacc) gln_bwd_means(d.pro_bacc + (size_t)b * SEP_STATS_SLOTS * 2, d.pro_stats + (size_t)b * SEP_STATS_SLOTS * 2, d.count, d.eps, mg, mgx);
        else { mg = d.pro_bsum[2 * b]; mgx = d.pro_bsum[2 * b + 1]; }
    }
    float dalpha_pro = 0.f;
    const bool writer = P_BWD && wr == 0 && rt == 0;           // the waves that own the store-back / slope gradient

    // This wave's share of one stage: 2 DMA instructions per operand (1 KiB each).  Every source address is a
    // wave-uniform base (SGPR pair, advanced with scalar adds: chunks are issued strictly in order) plus a per-lane
    // 32-bit offset fixed for the whole tile.
    const int Ks1 = SPLIT ? d.k_split : d.K;
    const int split_chunk = SPLIT ? d.k_split / DK : -1;
    const size_t stepB = (size_t)DK * d.ldt;
    const size_t stepA = TRANS_A ? (size_t)DK * d.M : (size_t)DK;
    const float* baseB[2];
    const float* baseC[2];
    const float* baseA[2];
    unsigned offA[2];                                             // BYTE offsets, unsigned: base(SGPR pair) + zext(voffset) is the
    const unsigned offB = 4u * (unsigned)(lk * d.ldt + 4 * l31);   // saddr form of global_load_lds (one VGPR, no 64-bit VALU add)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r2 = 4 * wid + 2 * q;                           // first of the two k rows this instruction covers
        baseB[q] = d.X + ((size_t)b * Ks1 + r2) * d.ldt + t0;
        baseC[q] = P_BWD ? d.pro_aux + ((size_t)b * Ks1 + r2) * d.ldt + t0 : nullptr;
        if (TRANS_A) {
            int mm = m0 + 4 * l31;
            if (mm > d.M - 4) mm = d.M - 4;                       // rows past M are never stored; keep the read in bounds
            baseA[q] = d.A + (size_t)r2 * d.M;
            offA[q] = 4u * (unsigned)(lk * d.M + mm);
        } else {
            // [128 m][16 k] image, 64-byte rows; instruction covers 16 rows; granule p of row r holds k-chunk p ^ ((r>>2)&3)
            const int r = lane >> 2, pch = lane & 3;
            int mm = m0 + 16 * (2 * wid + q) + r;
            if (mm > d.M - 1) mm = d.M - 1;
            baseA[q] = d.A;
            offA[q] = 4u * (unsigned)(mm * Ks1 + 4 * (pch ^ ((r >> 2) & 3)));
        }
    }
    int kci = 0;                                                  // next chunk to issue
    auto issue = [&](const int stage) {
        if (SPLIT && kci == split_chunk) {                        // uniform, taken once per tile: switch to the second source
            const int Ks2 = d.K - d.k_split;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r2 = 4 * wid + 2 * q;
                baseB[q] = d.X2 + ((size_t)b * Ks2 + r2) * d.ldt + t0;
                if (TRANS_A) baseA[q] = d.A2 + (size_t)r2 * d.M;
                else {
                    const int r = lane >> 2, pch = lane & 3;
                    int mm = m0 + 16 * (2 * wid + q) + r;
                    if (mm > d.M - 1) mm = d.M - 1;
                    baseA[q] = d.A2;
                    offA[q] = 4u * (unsigned)(mm * Ks2 + 4 * (pch ^ ((r >> 2) & 3)));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16_asm(baseB[q], offB, lds_addr(&sm.Bs[stage][(4 * wid + 2 * q) * 128]));
            if (P_BWD) glds16_asm(baseC[q], offB, lds_addr(&sm.Cs[stage][(4 * wid + 2 * q) * 128]));
            glds16_asm(baseA[q], offA[q], lds_addr(TRANS_A ? &sm.As[stage][(4 * wid + 2 * q) * 128] : &sm.As[stage][(2 * wid + q) * 256]));
            baseB[q] += stepB;
            if (P_BWD) baseC[q] += stepB;
            baseA[q] += stepA;
        }
        ++kci;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // Software pipeline at half-chunk granularity.  On a SIMD the co-resident waves share the matrix pipe fairly, so they
    // run in lockstep and reach their barriers / LDS reads TOGETHER: whatever a wave does between two MFMA bursts is pipe
    // idle time for all of them (measured: MFMA idle ~= SQ_WAIT_ANY + non-MFMA issue).  So each wave keeps its own MFMA
    // stream dense: the fragments of half h+1 are read from LDS while the 16 MFMAs of half h run, and the per-chunk
    // barrier sits between two bursts with the next burst's operands already in registers.  Same 32 fragment registers.
    float fa[2][2][4], fb[2][2][4];      // [half][mi|ni][kk]
    float fc[2][2][4];                   // GLN_BWD: the pre-activation fragments
    float fs[2][4], fh[2][4];            // gLN prologues: the per-row scale / shift of the half's four k
    auto read_half = [&](const int stage, const int h, const int kc) {
        const float* Ab = sm.As[stage];
        const float* Bb = sm.Bs[stage];
        if (TRANS_A) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fa[h][0][kk] = Ab[(8 * lk + 4 * h + kk) * 128 + wr * 64 + l31];
                fa[h][1][kk] = Ab[(8 * lk + 4 * h + kk) * 128 + wr * 64 + 32 + l31];
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = wr * 64 + mi * 32 + l31;
                const float* p = Ab + m * 16 + 4 * ((2 * lk + h) ^ ((m >> 2) & 3));
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[h][mi][e] = p[e];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fb[h][0][kk] = Bb[(8 * lk + 4 * h + kk) * 128 + wc * 64 + l31];
            fb[h][1][kk] = Bb[(8 * lk + 4 * h + kk) * 128 + wc * 64 + 32 + l31];
        }
        if (P_BWD) {
            const float* Cb = sm.Cs[stage];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fc[h][0][kk] = Cb[(8 * lk + 4 * h + kk) * 128 + wc * 64 + l31];
                fc[h][1][kk] = Cb[(8 * lk + 4 * h + kk) * 128 + wc * 64 + 32 + l31];
            }
        }
        if (P_GLN || P_BWD) {
            const int kbase = kc * DK + 8 * lk + 4 * h;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fs[h][kk] = sm.sc[kbase + kk];
                if (P_GLN) fh[h][kk] = sm.sh[kbase + kk];
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // the reads stay HERE: ahead of the MFMA burst that hides their latency
    };
    const bool live0 = t0 + wc * 64 + l31 < d.T, live1 = t0 + wc * 64 + 32 + l31 < d.T;
    const unsigned st_lane_off = 4u * (unsigned)(8 * lk * d.ldt + l31);      // GLN_BWD store-back: this lane's byte offset in the chunk
    auto apply_pro = [&](const int kc, const int h, const int kk) {
        if (P_BWD) {
            // d(pre-activation) = rstd*(gamma_k*dv - mg - xhat*mgx) * PReLU'(a)   on the fragments
            const float gk = fs[h][kk];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const bool live = ni ? live1 : live0;
                const float a = fc[h][ni][kk];
                const float u = prelu_f(a, alpha_p);
                const float xh = (u - mu) * rstd;
                const float du = rstd * (gk * fb[h][ni][kk] - mg - xh * mgx);
                const float da = live ? du * prelu_grad(a, alpha_p) : 0.f;
                fb[h][ni][kk] = da;
                if (writer) {
                    if (live && a <= 0.f) dalpha_pro += du * a;
                    // uniform row pointer (SGPR pair) + per-lane byte offset: the saddr form of global_store
                    float* srow = d.pro_store + ((size_t)b * d.K + kc * DK + 4 * h + kk) * d.ldt + t0 + wc * 64 + ni * 32;
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(srow) + (size_t)st_lane_off) = da;
                }
            }
        } else if (PRO != SEP_PRO_NONE) {
            const float scv = P_GLN ? fs[h][kk] : 1.f;
            const float shv = P_GLN ? fh[h][kk] : 0.f;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                float v = fb[h][ni][kk];
                if (P_PRELU) v = prelu_f(v, alpha_p);
                fb[h][ni][kk] = P_GLN ? v * scv + shv : v;
            }
        }
    };
    auto mfma_half = [&](const int kc, const int h) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            apply_pro(kc, h, kk);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][0][kk], fb[h][0][kk], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][0][kk], fb[h][1][kk], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][1][kk], fb[h][0][kk], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][1][kk], fb[h][1][kk], acc[1][1], 0, 0, 0);
        }
    };
    // AR == 1: whole-chunk steps.  prologue + split of chunk kc (VALU), then -- behind the barrier that says chunk kc+1 has
    // landed -- the LDS reads of chunk kc+1 in three portions between the four groups of six bf16 MFMAs of chunk kc, so
    // that no more than one portion of raw fragments is live beside the 48 packed operand registers.
    auto read_a6 = [&](const int stage) {
        const float* Ab = sm.As[stage];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (TRANS_A) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    fa[h][0][kk] = Ab[(8 * lk + 4 * h + kk) * 128 + wr * 64 + l31];
                    fa[h][1][kk] = Ab[(8 * lk + 4 * h + kk) * 128 + wr * 64 + 32 + l31];
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int m = wr * 64 + mi * 32 + l31;
                    const float* p = Ab + m * 16 + 4 * ((2 * lk + h) ^ ((m >> 2) & 3));
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[h][mi][e] = p[e];
                }
            }
        }
    };
    auto read_b6 = [&](const int stage, const int ni) {
        const float* Bb = sm.Bs[stage];
        const float* Cb = sm.Cs[P_BWD ? stage : 0];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fb[h][ni][kk] = Bb[(8 * lk + 4 * h + kk) * 128 + wc * 64 + ni * 32 + l31];
                if (P_BWD) fc[h][ni][kk] = Cb[(8 * lk + 4 * h + kk) * 128 + wc * 64 + ni * 32 + l31];
            }
    };
    auto step6 = [&](const int kc, const int stage, const int nstage) {
        u32x4_t pa[2][3], pb[2][3];
        if (P_GLN || P_BWD) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    fs[h][kk] = sm.sc[kc * DK + 8 * lk + 4 * h + kk];
                    if (P_GLN) fh[h][kk] = sm.sh[kc * DK + 8 * lk + 4 * h + kk];
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            split3_frag(fa[0][i], fa[1][i], pa[i]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) apply_pro(kc, h, kk);
#pragma unroll
        for (int i = 0; i < 2; ++i) split3_frag(fb[0][i], fb[1][i], pb[i]);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kc + 1 < nk;
        if (more) {
            // chunk kc+1 has landed (mine: all but the DMAs of chunk kc+2; everyone's: barrier), and every wave is past
            // its reads of chunk kc's stage, which chunk kc+NS now overwrites
            wait_keep4_and_barrier(NS == 3 && kc + 2 < nk);
            if (kc + NS < nk) issue(stage);
            read_a6(nstage);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_split6(pa[0], pb[0], acc[0][0]);
        mfma_split6(pa[1], pb[0], acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_b6(nstage, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_split6(pa[0], pb[1], acc[0][1]);
        mfma_split6(pa[1], pb[1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_b6(nstage, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    int bexp[2] = {0, 0};                // per-lane scale exponent of this lane's column in column block ni
    bool bset[2] = {false, false};       // ... chosen yet?  (a column stays unset while it has only seen zeros)
    int aexp = 0;                        // A * 2^aexp < 2^13
    if (AR == 2) aexp = 13 - __builtin_amdgcn_frexp_expf(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, d.a_amax[0]))));
    auto step3h = [&](const int kc, const int stage, const int nstage) {
        u32x4_t pa[2][2], pb[2][2];
        if (P_GLN) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    fs[h][kk] = sm.sc[kc * DK + 8 * lk + 4 * h + kk];
                    fh[h][kk] = sm.sh[kc * DK + 8 * lk + 4 * h + kk];
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fa[h][i][kk] = __builtin_ldexpf(fa[h][i][kk], aexp);
            split2_frag(fa[0][i], fa[1][i], pa[i]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) apply_pro(kc, h, kk);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float m = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) m = fmaxf(m, fabsf(fb[h][ni][kk]));
            m = fmaxf(m, __shfl_xor(m, 32, 64));                          // the other lane half holds the column's other 8 k
            const int e = __builtin_amdgcn_frexp_expf(m);                  // m = f * 2^e, f in [0.5, 1)
            const bool grow = m > 0.f && (!bset[ni] || e + bexp[ni] > 14);      // first non-zero chunk, or the column outgrew its scale
            bset[ni] = bset[ni] || m > 0.f;
            const int nexp = grow ? 9 - e : bexp[ni];
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {                  // rare after the first chunks
                const int delta = nexp - bexp[ni];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], delta);
            }
            bexp[ni] = nexp;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fb[h][ni][kk] = __builtin_ldexpf(fb[h][ni][kk], nexp);
            split2_frag(fb[0][ni], fb[1][ni], pb[ni]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kc + 1 < nk;
        if (more) {
            wait_keep4_and_barrier(NS == 3 && kc + 2 < nk);
            if (kc + NS < nk) issue(stage);
            read_a6(nstage);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_split3(pa[0], pb[0], acc[0][0]);
        mfma_split3(pa[1], pb[0], acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_b6(nstage, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_split3(pa[0], pb[1], acc[0][1]);
        mfma_split3(pa[1], pb[1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_b6(nstage, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](const int kc, const int stage) {
        read_half(stage, 1, kc);
        mfma_half(kc, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nk) {
            // chunk kc+1 has landed (mine: vmcnt(0); everyone's: barrier) and every wave is past its reads of this stage
            wait_all_and_barrier();
            if (kc + 2 < nk) issue(stage);
            read_half(stage ^ 1, 0, kc + 1);
        }
        mfma_half(kc, 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    PROF_STAMP(1);
    issue(0);
    if (NS == 3 && nk > 1) {
        issue(1);
        wait_keep4_and_barrier(true);
        if (nk > 2) issue(2);
    } else {
        wait_all_and_barrier();
        if (nk > 1) issue(1);
    }
    PROF_STAMP(2);
    if (AR == 0) read_half(0, 0, 0);
    int kc = 0;
    if (AR == 2) {
        read_a6(0);
        read_b6(0, 0);
        read_b6(0, 1);
        for (; kc + 2 < nk; kc += 3) {
            step3h(kc, 0, 1);
            step3h(kc + 1, 1, 2);
            step3h(kc + 2, 2, 0);
        }
        if (kc < nk) step3h(kc, 0, 1);
        if (kc + 1 < nk) step3h(kc + 1, 1, 2);
        // undo the scales: 2^aexp of A, 2^bexp of this lane's column
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], -aexp - bexp[ni]);
    } else if (AR == 1) {
        read_a6(0);
        read_b6(0, 0);
        read_b6(0, 1);
        if (NS == 3) {
            for (; kc + 2 < nk; kc += 3) {
                step6(kc, 0, 1);
                step6(kc + 1, 1, 2);
                step6(kc + 2, 2, 0);
            }
            if (kc < nk) step6(kc, 0, 1);
            if (kc + 1 < nk) step6(kc + 1, 1, 2);
        } else {
            for (; kc + 1 < nk; kc += 2) {
                step6(kc, 0, 1);
                step6(kc + 1, 1, 0);
            }
            if (kc < nk) step6(kc, 0, 1);
        }
    } else {
        for (; kc + 1 < nk; kc += 2) {       // two steps per trip so the stage index is a literal
            step(kc, 0);
            step(kc + 1, 1);
        }
        if (kc < nk) step(kc, 0);
    }
    PROF_STAMP(3);
    __syncthreads();
    PROF_STAMP(4);
    // Launder the thread id: everything the epilogue derives from it (lane offsets, row pointers) would otherwise be
    // hoisted above the main loop as loop-invariant and held in registers through it -- at 128 VGPRs that spilled INSIDE
    // the loop (a scratch reload waits on vmcnt, i.e. on the DMA ring).
    int etid = tid, eb = b, em0 = m0, et0 = t0;
    asm volatile("" : "+v"(etid), "+s"(eb), "+s"(em0), "+s"(et0));
    const int ewid = __builtin_amdgcn_readfirstlane(etid >> 6);
    const int elane = etid & 63;
    gemm_epilogue<EF>(d, acc, eb, em0, et0, ewid >> 1, ewid & 1, elane >> 5, elane & 31, etid, &sm.As[0][0], sm.red, BN, true PROF_PASS);
    PROF_STAMP(5);
    if (P_BWD && rt == 0) {
        const double sdal = block_sum_256<double>((double)dalpha_pro, sm.red);
        if (tid == 0) atomicAdd(d.pro_dalpha, sdal);
    }
#ifdef SEP_PROF
    __builtin_amdgcn_s_waitcnt(0x0070);
#endif
    PROF_STAMP(6);
#ifdef SEP_PROF
    if (tid == 0 && bid < 8192) g_blk_end[bid] = wall_clock64();
#endif
}

// ======================================================================================
// weight gradient: reduction over (batch, frame) columns, split into nsplit slabs
// ======================================================================================
constexpr int WK = 32;      // frames per chunk (one 128-byte line per operand row)
constexpr int LDW_S = 129;  // k-major LDS row stride: odd -> conflict-free transposing writes AND fragment reads

struct __attribute__((aligned(16))) WgradSmem {
    float Gs[2][WK * LDW_S];
    float Xs[2][WK * LDW_S];
};

__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(const sep_wgrad_desc d) {
    __shared__ WgradSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int ntiles = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int tile = j % ntiles;
    const int s = (j / ntiles) * 8 + xcd;
    if (s >= d.nsplit) return;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const int cps_t = d.ldt / WK;                 // chunks per sample
    const long chunks_total = (long)d.B * cps_t;
    const long cper = (chunks_total + d.nsplit - 1) / d.nsplit;
    const long c_begin = (long)s * cper;
    long c_end = c_begin + cper;
    if (c_end > chunks_total) c_end = chunks_total;

    const int lr = tid >> 3, lc4 = tid & 7;       // staging: rows lr + 32*i (i<4), float4 column lc4
    const bool do_bias = d.partial_bias != nullptr && (tile % ntn) == 0;
    float bias_acc = 0.f;

    const float alpha_x = (d.x_mode == SEP_PRO_PRELU || d.x_mode == SEP_PRO_GLN_PRELU) ? d.x_alpha[0] : 0.f;
    float4 rg[4], rx[4];
    float xsc[4], xsh[4];
    int stat_b = -1;            // sample whose gLN constants are cached below
    float stat_mu = 0.f, stat_rstd = 1.f;
    float xgam[4] = {0.f, 0.f, 0.f, 0.f}, xbet[4] = {0.f, 0.f, 0.f, 0.f};   // rows are fixed per thread: fetch once
    if (d.x_mode == SEP_PRO_GLN || d.x_mode == SEP_PRO_GLN_PRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + lr + 32 * i;
            if (n < d.N) { xgam[i] = d.x_gamma[n]; xbet[i] = d.x_beta[n]; }
        }
    }

    auto chunk_valid = [&](long c) -> bool { return (int)(c % cps_t) * WK < d.T; };

    auto load_global = [&](long c) {
        const int b = (int)(c / cps_t);
        const int t0 = (int)(c % cps_t) * WK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lr + 32 * i;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < d.M) {
                const float* src;
                size_t off;
                if (d.g_split && m >= d.g_split) { src = d.G2; off = ((size_t)b * (d.M - d.g_split) + (m - d.g_split)) * d.ldt; }
                else { src = d.G; off = ((size_t)b * (d.g_split ? d.g_split : d.M) + m) * d.ldt; }
                g = ld4(src + off + t0 + 4 * lc4);
                if (d.g_mul) {
                    const float4 w = ld4(d.Gaux + ((size_t)(b / d.g_div) * d.M + m) * d.ldt + t0 + 4 * lc4);
                    g.x *= w.x; g.y *= w.y; g.z *= w.z; g.w *= w.w;
                }
            }
            rg[i] = g;
            const int n = n0 + lr + 32 * i;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            xsc[i] = 0.f; xsh[i] = 0.f;
            if (n < d.N) {
                const int bx = b / d.x_div;
                x = ld4(d.X + ((size_t)bx * d.N + n) * d.ldt + t0 + 4 * lc4);
                if (d.x_mode == SEP_PRO_GLN || d.x_mode == SEP_PRO_GLN_PRELU) {
                    if (bx != stat_b) { gln_mu_rstd(d.x_stats + (size_t)bx * SEP_STATS_SLOTS * 2, d.count, d.eps, stat_mu, stat_rstd); stat_b = bx; }
                    xsc[i] = xgam[i] * stat_rstd;
                    xsh[i] = xbet[i] - stat_mu * xsc[i];
                }
            }
            rx[i] = x;
        }
    };

    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mm = lr + 32 * i;
            sm.Gs[buf][(4 * lc4 + 0) * LDW_S + mm] = rg[i].x;
            sm.Gs[buf][(4 * lc4 + 1) * LDW_S + mm] = rg[i].y;
            sm.Gs[buf][(4 * lc4 + 2) * LDW_S + mm] = rg[i].z;
            sm.Gs[buf][(4 * lc4 + 3) * LDW_S + mm] = rg[i].w;
            float v[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
            const bool row_ok = (n0 + mm) < d.N;
            if (d.x_mode == SEP_PRO_PRELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = prelu_f(v[q], alpha_x);
            } else if (d.x_mode == SEP_PRO_GLN || d.x_mode == SEP_PRO_GLN_PRELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = (d.x_mode == SEP_PRO_GLN_PRELU) ? prelu_f(v[q], alpha_x) : v[q];
                    v[q] = row_ok ? (u * xsc[i] + xsh[i]) : 0.f;
                }
            }
            sm.Xs[buf][(4 * lc4 + 0) * LDW_S + mm] = v[0];
            sm.Xs[buf][(4 * lc4 + 1) * LDW_S + mm] = v[1];
            sm.Xs[buf][(4 * lc4 + 2) * LDW_S + mm] = v[2];
            sm.Xs[buf][(4 * lc4 + 3) * LDW_S + mm] = v[3];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // skip chunks that lie entirely in the zero pad (G is exactly 0 there)
    long c = c_begin;
    while (c < c_end && !chunk_valid(c)) ++c;
    const int lk = lane >> 5, l31 = lane & 31;
    int cur = 0;
    if (c < c_end) {
        load_global(c);
        store_lds(0);
    }
    __syncthreads();
    while (c < c_end) {
        long cn = c + 1;
        while (cn < c_end && !chunk_valid(cn)) ++cn;
        if (cn < c_end) load_global(cn);
        mfma_chunk32(sm.Gs[cur], LDW_S, sm.Xs[cur], LDW_S, wr * 64 + l31, wc * 64 + l31, lk, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        if (do_bias && tid < BM) {
            float sacc = 0.f;
#pragma unroll 8
            for (int k = 0; k < WK; ++k) sacc += sm.Gs[cur][k * LDW_S + tid];
            bias_acc += sacc;
        }
        if (cn < c_end) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        c = cn;
    }

    auto put_tile = [&](auto atomic_c) {                // a plain store into slab s
        constexpr bool AT = decltype(atomic_c)::value;
        float* out = d.partial + (AT ? 0 : (size_t)s * d.M * d.N);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < d.M) {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int col = n0 + wc * 64 + ni * 32 + l31;
                        if (col < d.N) wg_put<AT>(out + (size_t)row * d.N + col, acc[mi][ni][r]);
                    }
                }
            }
        if (do_bias && tid < BM && (m0 + tid) < d.M) wg_put<AT>(d.partial_bias + (AT ? 0 : (size_t)s * d.M) + m0 + tid, bias_acc);
    };
    put_tile(std::false_type{});
}


// ======================================================================================
// Direct-to-LDS weight gradient:  partial[s][m][n] = sum over slab s of G[b][m][t] * pro(X[b][n][t]).
// Both operands are row-major with the contraction index (frame t) contiguous, i.e. both take the "row-major A"
// route of pw_gemm_direct_kernel: 16-frame chunks, 64-byte rows DMA'd with the XOR swizzle on the source address,
// fragments by one ds_read_b128 per 32-row block and half-chunk.
//
// Occupancy without more slabs: a slab's partial costs M*N*4 bytes of HBM write + re-read, so doubling the slabs to fill
// the chip would add ~3 GB per step.  Instead a workgroup has EIGHT waves = two 4-wave groups that own alternate chunks of
// the same slab (intra-block split of the contraction), each group a self-contained 2-stage ring + half-chunk software
// pipeline exactly like the GEMM above; the groups share the per-chunk barrier, and group 1 hands its accumulators to
// group 0 through LDS (the 64 KiB of the two rings) at the end.  2 workgroups per CU = 4 waves per SIMD, <= 128 VGPRs.
// The gLN / PReLU prologue of X acts on the B fragments (per-lane row affine x per-sample mean/rstd table in LDS); bias
// row sums fall out of the A fragments.  Frames >= T need no mask: G is zero there by the layout contract.
// ======================================================================================

constexpr int WMAXB = 256;      // samples whose gLN constants fit the LDS table

struct __attribute__((aligned(16))) WDirectSmem {
    float Gs[2][2][128 * DK];   // [group][stage]
    float Xs[2][2][128 * DK];
    float mu[WMAXB];
    float rstd[WMAXB];
};

template <int XMODE>
__global__ __launch_bounds__(512, 4) void pw_wgrad_direct_kernel(const sep_wgrad_desc d) {
    constexpr bool X_GLN = XMODE == SEP_PRO_GLN || XMODE == SEP_PRO_GLN_PRELU;
    constexpr bool X_PRELU = XMODE == SEP_PRO_PRELU || XMODE == SEP_PRO_GLN_PRELU;
    __shared__ WDirectSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid8 >> 2, wid = wid8 & 3;
    const int wr = wid >> 1, wc = wid & 1;
    const int lk = lane >> 5, l31 = lane & 31;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int ntiles = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int tile = j % ntiles;
    const int s = (j / ntiles) * 8 + xcd;
    if (s >= d.nsplit) return;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const int cps_t = d.ldt / DK;                  // chunks per sample
    const long chunks_total = (long)d.B * cps_t;
    const long cper = (chunks_total + d.nsplit - 1) / d.nsplit;
    const long c_begin = (long)s * cper;
    long c_end = c_begin + cper;
    if (c_end > chunks_total) c_end = chunks_total;
    const int nk_all = (int)(c_end > c_begin ? c_end - c_begin : 0);
    const int nk_max = (nk_all + 1) >> 1;          // trips of the shared loop (group 0's chunk count)
    const int nk = (nk_all + 1 - grp) >> 1;        // chunks of THIS group: c_begin + grp, +2, ...

    const float alpha_x = X_PRELU ? d.x_alpha[0] : 0.f;
    if (X_GLN) {
        const int nb = d.B / d.x_div;
        for (int bx = tid; bx < nb; bx += 512) {
            float mu, rstd;
            gln_mu_rstd(d.x_stats + (size_t)bx * SEP_STATS_SLOTS * 2, d.count, d.eps, mu, rstd);
            sm.mu[bx] = mu; sm.rstd[bx] = rstd;
        }
    }
    // this lane's two X rows (B operand) and their gLN affine
    float xg[2] = {0.f, 0.f}, xb[2] = {0.f, 0.f};
    if (X_GLN) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wc * 64 + ni * 32 + l31;
            if (n < d.N) { xg[ni] = d.x_gamma[n]; xb[ni] = d.x_beta[n]; }
        }
    }
    // consume the loads NOW: the compiler does not count the asm LDS-DMAs below, so a wait it placed at a first use inside
    // the loop would be vmcnt(0) and drain the ring
    asm volatile("" :: "v"(xg[0]), "v"(xg[1]), "v"(xb[0]), "v"(xb[1]), "v"(alpha_x));

    // G source of this row tile (g_split is a multiple of BM -> block-uniform)
    const bool gsecond = d.g_split && m0 >= d.g_split;
    const float* Gsrc = gsecond ? d.G2 : d.G;
    const int Mg = gsecond ? d.M - d.g_split : (d.g_split ? d.g_split : d.M);
    const int mg0 = gsecond ? m0 - d.g_split : m0;
    const int Mg_lim = (d.M < m0 + BM ? d.M : m0 + BM) - (gsecond ? d.g_split : 0);   // rows of this tile that exist in Gsrc

    // DMA sources: wave-uniform base of (sample, first frame) + this lane's fixed byte offset (row, swizzled 16-byte granule)
    const int r16 = lane >> 2, cch = (lane & 3) ^ ((r16 >> 2) & 3);
    unsigned voffG[2], voffX[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g16 = 2 * wid + q;                              // 16-row group of the 128-row tile
        int mm = mg0 + 16 * g16 + r16;
        if (mm > Mg_lim - 1) mm = Mg_lim - 1;                     // rows past M: in-bounds garbage, never stored
        voffG[q] = 4u * (unsigned)(mm * d.ldt + 4 * cch);
        int nn = n0 + 16 * g16 + r16;
        if (nn > d.N - 1) nn = d.N - 1;
        voffX[q] = 4u * (unsigned)(nn * d.ldt + 4 * cch);
    }
    // (sample, frame) of the next chunk this group issues, and of the chunk it computes
    const long c_first = c_begin + grp;
    int ib = (int)(c_first / cps_t), it = (int)(c_first % cps_t);
    int ibx = ib / d.x_div, ibm = ib % d.x_div;
    int cb = ib, ct = it, cbx = ibx, cbm = ibm;
    auto issue = [&](const int stage) {
        const float* bG = Gsrc + (size_t)ib * Mg * d.ldt + it * DK;
        const float* bX = d.X + (size_t)ibx * d.N * d.ldt + it * DK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16_asm(bG, voffG[q], lds_addr(&sm.Gs[grp][stage][(2 * wid + q) * 256]));
            glds16_asm(bX, voffX[q], lds_addr(&sm.Xs[grp][stage][(2 * wid + q) * 256]));
        }
        it += 2;
        if (it >= cps_t) {
            it -= cps_t; ++ib;
            if (++ibm == d.x_div) { ibm = 0; ++ibx; }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    float bias_acc[2] = {0.f, 0.f};
    const bool do_bias = d.partial_bias != nullptr && (tile % ntn) == 0 && wc == 0;

    float fa[2][2][4], fb[2][2][4];      // [half][mi|ni][frame]
    auto read_half = [&](const int stage, const int h) {
        const float* Gb = sm.Gs[grp][stage];
        const float* Xb = sm.Xs[grp][stage];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m = wr * 64 + mi * 32 + l31;
            const float* p = Gb + m * 16 + 4 * ((2 * lk + h) ^ ((m >> 2) & 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) fa[h][mi][e] = p[e];
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = wc * 64 + ni * 32 + l31;
            const float* p = Xb + n * 16 + 4 * ((2 * lk + h) ^ ((n >> 2) & 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) fb[h][ni][e] = p[e];
        }
        __builtin_amdgcn_sched_barrier(0);      // the reads stay HERE: ahead of the MFMA burst that hides their latency
    };
    float scv[2] = {1.f, 1.f}, shv[2] = {0.f, 0.f};
    auto set_sample = [&]() {
        if (X_GLN) {
            const float mu = sm.mu[cbx], rstd = sm.rstd[cbx];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) { scv[ni] = xg[ni] * rstd; shv[ni] = xb[ni] - mu * scv[ni]; }
        }
    };
    auto mfma_half = [&](const int h) {
        if (do_bias) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) bias_acc[mi] += fa[h][mi][kk];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (XMODE != SEP_PRO_NONE) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    float v = fb[h][ni][kk];
                    if (X_PRELU) v = prelu_f(v, alpha_x);
                    fb[h][ni][kk] = X_GLN ? v * scv[ni] + shv[ni] : v;
                }
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][0][kk], fb[h][0][kk], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][0][kk], fb[h][1][kk], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][1][kk], fb[h][0][kk], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][1][kk], fb[h][1][kk], acc[1][1], 0, 0, 0);
        }
    };

#ifdef SEP_PROF
    const int prof_slot = (bid == 8 ? 0 : bid == 200 ? 1 : bid == 201 ? 2 : bid == (int)gridDim.x - 9 ? 3 : -1);
#define WPROF(k) do { if (prof_slot >= 0 && lane == 0 && grp == 0) g_prof[prof_slot][wid][k] = clock64(); } while (0)
#else
#define WPROF(k) do { } while (0)
#endif
    WPROF(0);
    __syncthreads();                                              // mu/rstd table visible; BEFORE the first DMA so it drains nothing
    set_sample();
    WPROF(1);
    if (nk > 0) issue(0);
    wait_all_and_barrier();
    WPROF(2);
    if (nk > 1) issue(1);
    if (nk > 0) read_half(0, 0);
    for (int i = 0; i < nk_max; ++i) {
        const int stage = i & 1;
        const bool act = i < nk;                                  // group 1 may own one chunk less: it still meets the barriers
        if (act) {
            read_half(stage, 1);
            mfma_half(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < nk_max) {
            // chunk i+1 has landed (mine: vmcnt(0); everyone's: barrier) and every wave is past its reads of this stage
            wait_all_and_barrier();
            if (i + 2 < nk) issue(stage);
            if (i + 1 < nk) read_half(stage ^ 1, 0);
        }
        if (act) mfma_half(1);
        __builtin_amdgcn_sched_barrier(0);
        // the chunk just computed was (cb, ct); this group's next one is two chunks on
        ct += 2;
        if (ct >= cps_t) {
            ct -= cps_t; ++cb;
            if (++cbm == d.x_div) { cbm = 0; ++cbx; }
            if (i + 1 < nk) set_sample();
        }
    }

    // The two groups hold partial sums of the same 128x128 tile.  Each keeps one 32-row half per wave (group g: mi = g),
    // hands the other half to its partner through LDS (the rings are dead now) and stores its own: all eight waves
    // share the exchange and the stores (first version: group 0 did everything, 34 k cycles of a 150 k-cycle workgroup).
    WPROF(3);
    __syncthreads();
    WPROF(4);
    float* red = &sm.Gs[0][0][0];                                 // 2 x 8192 floats = 32 handed accumulators x 256 lanes per group
    const int tg = tid & 255;
    auto hand_over = [&](f32x16 (&give)[2], const float bias_give, float* bias_slot) {
        float* mine = red + grp * 8192;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(ni * 16 + r) * 256 + tg] = give[ni][r];
        bias_slot[tg] = bias_give;
    };
    if (grp == 0) hand_over(acc[1], bias_acc[1], sm.mu);           // wave-uniform branch, not 64 selects
    else hand_over(acc[0], bias_acc[0], sm.rstd);
    __syncthreads();
    WPROF(5);
    auto finish = [&](f32x16 (&keep)[2], const float bias_keep, const float* bias_slot, const int mi) {
        const float* theirs = red + (1 - grp) * 8192;
        auto put_tile = [&](auto atomic_c) {
            constexpr bool AT = decltype(atomic_c)::value;
            float* out = d.partial + (AT ? 0 : (size_t)s * d.M * d.N);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col = n0 + wc * 64 + ni * 32 + l31;
                    const float v = keep[ni][r] + theirs[(ni * 16 + r) * 256 + tg];
                    if (row < d.M && col < d.N) wg_put<AT>(out + (size_t)row * d.N + col, v);
                }
            }
            if (do_bias) {
                const float mine_b = bias_keep + bias_slot[tg];
                const float tot = mine_b + __shfl_xor(mine_b, 32, 64);     // the two lane halves own different frames
                const int row = m0 + wr * 64 + mi * 32 + l31;
                if (lk == 0 && row < d.M) wg_put<AT>(d.partial_bias + (AT ? 0 : (size_t)s * d.M) + row, tot);
            }
        };
        put_tile(std::false_type{});
    };
    if (grp == 0) finish(acc[0], bias_acc[0], sm.rstd, 0);
    else finish(acc[1], bias_acc[1], sm.mu, 1);
#ifdef SEP_PROF
    __builtin_amdgcn_s_waitcnt(0x0070);
#endif
    WPROF(6);
#undef WPROF
}

// ======================================================================================
// Weight gradient in the split arithmetic (SEP_ARITH_BF16X6, see pw_gemm_direct_kernel): the same tiles, DMA layout and
// fragment reads as pw_wgrad_direct_kernel, but 4-wave workgroups that own a whole slab (no in-block split of the
// contraction: the packed operands need 168 registers, so three independent workgroups per CU give the overlap that two
// 8-wave ones gave), a 3-stage ring, whole-chunk steps: prologue + split of chunk i (VALU), barrier, then the LDS reads
// of chunk i+1 in three portions between the four groups of six bf16 MFMAs.  Same slab count as the 8-wave kernel.
// ======================================================================================
struct __attribute__((aligned(16))) WSplitSmem {
    float Gs[3][128 * DK];
    float Xs[3][128 * DK];
    float mu[WMAXB];
    float rstd[WMAXB];
};

template <int XMODE>
__global__ __launch_bounds__(256, 3) void pw_wgrad_split_kernel(const sep_wgrad_desc d) {
    constexpr bool X_GLN = XMODE == SEP_PRO_GLN || XMODE == SEP_PRO_GLN_PRELU;
    constexpr bool X_PRELU = XMODE == SEP_PRO_PRELU || XMODE == SEP_PRO_GLN_PRELU;
    constexpr int NS = 3;
    __shared__ WSplitSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int lk = lane >> 5, l31 = lane & 31;
    const int ntm = (d.M + BM - 1) / BM, ntn = (d.N + BN - 1) / BN;
    const int ntiles = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int tile = j % ntiles;
    const int s = (j / ntiles) * 8 + xcd;
    if (s >= d.nsplit) return;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const int cps_t = d.ldt / DK;                  // chunks per sample
    const long chunks_total = (long)d.B * cps_t;
    const long cper = (chunks_total + d.nsplit - 1) / d.nsplit;
    const long c_begin = (long)s * cper;
    long c_end = c_begin + cper;
    if (c_end > chunks_total) c_end = chunks_total;
    const int nk = (int)(c_end > c_begin ? c_end - c_begin : 0);

    const float alpha_x = X_PRELU ? d.x_alpha[0] : 0.f;
    if (X_GLN) {
        const int nb = d.B / d.x_div;
        for (int bx = tid; bx < nb; bx += 256) {
            float mu, rstd;
            gln_mu_rstd(d.x_stats + (size_t)bx * SEP_STATS_SLOTS * 2, d.count, d.eps, mu, rstd);
            sm.mu[bx] = mu; sm.rstd[bx] = rstd;
        }
    }
    float xg[2] = {0.f, 0.f}, xb[2] = {0.f, 0.f};
    if (X_GLN) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wc * 64 + ni * 32 + l31;
            if (n < d.N) { xg[ni] = d.x_gamma[n]; xb[ni] = d.x_beta[n]; }
        }
    }
    asm volatile("" :: "v"(xg[0]), "v"(xg[1]), "v"(xb[0]), "v"(xb[1]), "v"(alpha_x));      // consumed before the first asm DMA

    const bool gsecond = d.g_split && m0 >= d.g_split;
    const float* Gsrc = gsecond ? d.G2 : d.G;
    const int Mg = gsecond ? d.M - d.g_split : (d.g_split ? d.g_split : d.M);
    const int mg0 = gsecond ? m0 - d.g_split : m0;
    const int Mg_lim = (d.M < m0 + BM ? d.M : m0 + BM) - (gsecond ? d.g_split : 0);

    const int r16 = lane >> 2, cch = (lane & 3) ^ ((r16 >> 2) & 3);
    unsigned voffG[2], voffX[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g16 = 2 * wid + q;
        int mm = mg0 + 16 * g16 + r16;
        if (mm > Mg_lim - 1) mm = Mg_lim - 1;
        voffG[q] = 4u * (unsigned)(mm * d.ldt + 4 * cch);
        int nn = n0 + 16 * g16 + r16;
        if (nn > d.N - 1) nn = d.N - 1;
        voffX[q] = 4u * (unsigned)(nn * d.ldt + 4 * cch);
    }
    int ib = (int)(c_begin / cps_t), it = (int)(c_begin % cps_t);
    int ibx = ib / d.x_div, ibm = ib % d.x_div;
    int ct = it, cbx = ibx, cbm = ibm;
    auto issue = [&](const int stage) {
        const float* bG = Gsrc + (size_t)ib * Mg * d.ldt + it * DK;
        const float* bX = d.X + (size_t)ibx * d.N * d.ldt + it * DK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16_asm(bG, voffG[q], lds_addr(&sm.Gs[stage][(2 * wid + q) * 256]));
            glds16_asm(bX, voffX[q], lds_addr(&sm.Xs[stage][(2 * wid + q) * 256]));
        }
        if (++it >= cps_t) {
            it = 0; ++ib;
            if (++ibm == d.x_div) { ibm = 0; ++ibx; }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    float bias_acc[2] = {0.f, 0.f};
    const bool do_bias = d.partial_bias != nullptr && (tile % ntn) == 0 && wc == 0;

    float fa[2][2][4], fb[2][2][4];      // [half][mi|ni][frame]
    auto read_g = [&](const int stage) {
        const float* Gb = sm.Gs[stage];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = wr * 64 + mi * 32 + l31;
                const float* p = Gb + m * 16 + 4 * ((2 * lk + h) ^ ((m >> 2) & 3));
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[h][mi][e] = p[e];
            }
    };
    auto read_x = [&](const int stage, const int ni) {
        const float* Xb = sm.Xs[stage];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = wc * 64 + ni * 32 + l31;
            const float* p = Xb + n * 16 + 4 * ((2 * lk + h) ^ ((n >> 2) & 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) fb[h][ni][e] = p[e];
        }
    };
    float s_mu = 0.f, s_rstd = 1.f;      // the current sample's statistics, wave-uniform (SGPRs)
    auto set_sample = [&]() {
        if (X_GLN) {
            s_mu = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sm.mu[cbx])));
            s_rstd = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sm.rstd[cbx])));
        }
    };
    auto step = [&](const int i, const int stage, const int nstage) {
        u32x4_t pa[2][3], pb[2][3];
        if (do_bias) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) bias_acc[mi] += fa[h][mi][kk];
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) split3_frag(fa[0][mi], fa[1][mi], pa[mi]);
        if (XMODE != SEP_PRO_NONE) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float scv = xg[ni] * s_rstd, shv = xb[ni] - s_mu * scv;      // re-formed per chunk: four registers fewer across the MFMAs
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        float v = fb[h][ni][kk];
                        if (X_PRELU) v = prelu_f(v, alpha_x);
                        fb[h][ni][kk] = X_GLN ? v * scv + shv : v;
                    }
            }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) split3_frag(fb[0][ni], fb[1][ni], pb[ni]);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = i + 1 < nk;
        if (more) {
            // chunk i+1 has landed (mine: all but the DMAs of chunk i+2; everyone's: barrier); every wave holds chunk i in
            // registers, so chunk i+3 may overwrite its stage
            wait_keep4_and_barrier(i + 2 < nk);
            if (i + NS < nk) issue(stage);
            read_g(nstage);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_split6(pa[0], pb[0], acc[0][0]);
        mfma_split6(pa[1], pb[0], acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_x(nstage, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_split6(pa[0], pb[1], acc[0][1]);
        mfma_split6(pa[1], pb[1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_x(nstage, 1);
        __builtin_amdgcn_sched_barrier(0);
        // the chunk just computed was frame chunk ct of sample cbx
        if (++ct >= cps_t) {
            ct = 0;
            if (++cbm == d.x_div) { cbm = 0; ++cbx; }
            if (more) set_sample();
        }
    };

    __syncthreads();                                              // mu/rstd table visible; before the first DMA so it drains nothing
    set_sample();
    if (nk > 0) {
        issue(0);
        if (nk > 1) {
            issue(1);
            wait_keep4_and_barrier(true);
            if (nk > 2) issue(2);
        } else
            wait_all_and_barrier();
        read_g(0);
        read_x(0, 0);
        read_x(0, 1);
        int i = 0;
        for (; i + 2 < nk; i += 3) {
            step(i, 0, 1);
            step(i + 1, 1, 2);
            step(i + 2, 2, 0);
        }
        if (i < nk) step(i, 0, 1);
        if (i + 1 < nk) step(i + 1, 1, 2);
    }

    // launder what the stores derive their addresses from: hoisted above the loop it would be held (and spilled) through it
    int etid = tid, es = s, em0 = m0, en0 = n0;
    asm volatile("" : "+v"(etid), "+s"(es), "+s"(em0), "+s"(en0));
    const int ewid = __builtin_amdgcn_readfirstlane(etid >> 6);
    const int ewr = ewid >> 1, ewc = ewid & 1, elk = (etid >> 5) & 1, el31 = etid & 31;
    auto put_tile = [&](auto atomic_c) {
        constexpr bool AT = decltype(atomic_c)::value;
        float* out = d.partial + (AT ? 0 : (size_t)es * d.M * d.N);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = em0 + ewr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * elk;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col = en0 + ewc * 64 + ni * 32 + el31;
                    if (row < d.M && col < d.N) wg_put<AT>(out + (size_t)row * d.N + col, acc[mi][ni][r]);
                }
            }
        if (do_bias) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float tot = bias_acc[mi] + __shfl_xor(bias_acc[mi], 32, 64);     // the two lane halves own different frames
                const int row = em0 + ewr * 64 + mi * 32 + el31;
                if (elk == 0 && row < d.M) wg_put<AT>(d.partial_bias + (AT ? 0 : (size_t)es * d.M) + row, tot);
            }
        }
    };
    put_tile(std::false_type{});
}

// ======================================================================================
constexpr int RMAXSEG = 64;
struct ReduceArgs {
    sep_reduce_seg seg[RMAXSEG];
    int blk_start[RMAXSEG + 1];
    unsigned char phased[RMAXSEG];      // 1: 64 outputs per workgroup, four slab phases per output
    int nseg;
};

// Two forms per segment, both with a fixed summation order (bit-stable run to run).  Launches that stream (the Conv-TasNet step's batched weight
// gradients: dozens of segments of 64 - 128 slabs of 64 K ... 128 K values, 0.5 GB per launch) keep one thread per output and four chains.  Short ones -- the
// dual-path models' dense layers: 170 slabs of 12 K values took 20 us, latency, 54 launches per DPTNet step -- get 64 outputs per
// workgroup and four threads per output: thread (i, g) adds the slabs k = g, g + 4, ... in four chains of its own (sixteen loads in flight
// per output), the four phases meet in LDS.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const ReduceArgs a) {
    __shared__ float ph[4][64];
    int sgi = 0;
    while (sgi + 1 < a.nseg && (int)blockIdx.x >= a.blk_start[sgi + 1]) ++sgi;
    const sep_reduce_seg sg = a.seg[sgi];
    if (!a.phased[sgi]) {
        const int i = ((int)blockIdx.x - a.blk_start[sgi]) * 256 + threadIdx.x;
        if (i >= sg.n) return;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < sg.nslab; k += 4) {
            s0 += sg.src[(size_t)(k + 0) * sg.stride + i];
            s1 += sg.src[(size_t)(k + 1) * sg.stride + i];
            s2 += sg.src[(size_t)(k + 2) * sg.stride + i];
            s3 += sg.src[(size_t)(k + 3) * sg.stride + i];
        }
        for (; k < sg.nslab; ++k) s0 += sg.src[(size_t)k * sg.stride + i];
        float v = ((s0 + s1) + (s2 + s3)) * sg.scale;
        if (sg.accumulate) v += sg.dst[i];
        sg.dst[i] = v;
        return;
    }
    const int g = threadIdx.x >> 6, li = threadIdx.x & 63;
    const int i = ((int)blockIdx.x - a.blk_start[sgi]) * 64 + li;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < sg.n) {
        int k = g;
        for (; k + 12 < sg.nslab; k += 16) {
            s0 += sg.src[(size_t)(k + 0) * sg.stride + i];
            s1 += sg.src[(size_t)(k + 4) * sg.stride + i];
            s2 += sg.src[(size_t)(k + 8) * sg.stride + i];
            s3 += sg.src[(size_t)(k + 12) * sg.stride + i];
        }
        for (; k < sg.nslab; k += 4) s0 += sg.src[(size_t)k * sg.stride + i];
    }
    ph[g][li] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < sg.n) {
        float v = ((ph[0][li] + ph[1][li]) + (ph[2][li] + ph[3][li])) * sg.scale;
        if (sg.accumulate) v += sg.dst[i];
        sg.dst[i] = v;
    }
}

__global__ void f64_to_f32_kernel(const double* src, float* dst, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (accumulate ? dst[i] : 0.f) + (float)src[i];
}

}  // namespace

#ifdef SEP_PROF
extern "C" int sep_debug_blocks(long long* start, long long* end) {
    if (hipMemcpyFromSymbol(start, HIP_SYMBOL(g_blk_start), sizeof(long long) * 8192) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(end, HIP_SYMBOL(g_blk_end), sizeof(long long) * 8192) == hipSuccess ? 0 : -1;
}
extern "C" int sep_debug_prof(long long* out) {      // development builds only (tools/gemm_prof.py)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 4 * 4 * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int sep_pw_gemm(const sep_gemm_desc* d, sep_stream_t stream) {
    SEP_REQUIRE(d != nullptr, "sep_pw_gemm: null descriptor");
    SEP_REQUIRE(d->B > 0 && d->M > 0 && d->K > 0 && d->T > 0, "sep_pw_gemm: empty problem (B=%d M=%d K=%d T=%d)", d->B, d->M, d->K, d->T);
    SEP_REQUIRE(d->ldt % 128 == 0 && d->ldt >= d->T, "sep_pw_gemm: ldt=%d must be a multiple of 128 and >= T=%d", d->ldt, d->T);
    SEP_REQUIRE(d->K % 16 == 0, "sep_pw_gemm: K=%d must be a multiple of 16", d->K);
    SEP_REQUIRE(d->k_split % 16 == 0 && d->k_split < d->K, "sep_pw_gemm: bad k_split=%d", d->k_split);
    SEP_REQUIRE(d->m_split % BM == 0 && d->m_split < d->M, "sep_pw_gemm: bad m_split=%d (M=%d)", d->m_split, d->M);
    SEP_REQUIRE(!d->trans_a || d->M % 4 == 0, "sep_pw_gemm: transposed A needs M %% 4 == 0 (M=%d)", d->M);
    SEP_REQUIRE(d->A && d->X && d->Y, "sep_pw_gemm: null operand");
    SEP_REQUIRE(!d->k_split || (d->A2 && d->X2), "sep_pw_gemm: k_split without A2/X2");
    SEP_REQUIRE(!d->m_split || d->Y2, "sep_pw_gemm: m_split without Y2");
    SEP_REQUIRE(d->pro_mode >= 0 && d->pro_mode <= SEP_PRO_GLN_BWD, "sep_pw_gemm: bad pro_mode %d", d->pro_mode);
    SEP_REQUIRE(d->arith >= SEP_ARITH_F32 && d->arith <= SEP_ARITH_F16X3, "sep_pw_gemm: bad arith %d", d->arith);
    if (d->pro_mode == SEP_PRO_PRELU || d->pro_mode == SEP_PRO_GLN_PRELU || d->pro_mode == SEP_PRO_GLN_BWD)
        SEP_REQUIRE(d->pro_alpha, "sep_pw_gemm: prologue needs pro_alpha");
    if (d->pro_mode >= SEP_PRO_GLN)
        SEP_REQUIRE(d->pro_stats && d->pro_gamma && (d->pro_mode == SEP_PRO_GLN_BWD || d->pro_beta) && d->count > 0, "sep_pw_gemm: gLN prologue needs stats/gamma/beta/count");
    if (d->pro_mode == SEP_PRO_GLN_BWD)
        SEP_REQUIRE(d->pro_aux && (d->pro_bsum || d->pro_bacc) && d->pro_store && d->pro_dalpha && !d->k_split, "sep_pw_gemm: GLN_BWD prologue needs aux/bsum/store/dalpha");
    if (d->pro_mode == SEP_PRO_GLN_BWD)      // row tile 0 stores da while the other row tiles still read X: in place only with one row tile
        SEP_REQUIRE(d->pro_store != d->X || d->M <= BM, "sep_pw_gemm: pro_store may alias X only when M <= %d (one row tile)", BM);
    if (d->epi_flags & SEP_EPI_STATS_PRELU) SEP_REQUIRE(d->epi_stats && d->epi_alpha, "sep_pw_gemm: STATS_PRELU needs epi_stats/epi_alpha");
    if (d->epi_flags & SEP_EPI_RESIDUAL) SEP_REQUIRE(d->epi_res, "sep_pw_gemm: RESIDUAL needs epi_res");
    if (d->epi_flags & SEP_EPI_PRELU_BWD) SEP_REQUIRE(d->epi_aux && d->epi_alpha && d->epi_dalpha, "sep_pw_gemm: PRELU_BWD needs aux/alpha/dalpha");
    if (d->epi_flags & SEP_EPI_ROWSUMS) SEP_REQUIRE(d->epi_aux && d->epi_rowpart && !d->m_split, "sep_pw_gemm: ROWSUMS needs aux/rowpart");
    if (d->epi_flags & SEP_EPI_ROWSUMS_PRELU) SEP_REQUIRE(d->epi_alpha, "sep_pw_gemm: ROWSUMS_PRELU needs epi_alpha");
    if (d->arith == SEP_ARITH_F16X3 && d->A_pk != nullptr) {
        SEP_REQUIRE(d->a_rscale != nullptr, "sep_pw_gemm: A_pk without a_rscale");
        if (sep_pw_gemm_packed(d, (hipStream_t)stream)) {      // packed weights, cooperative split (gemm_coop.hip)
            SEP_CHECK_LAUNCH("sep_pw_gemm (packed)");
            return 0;
        }
    }
    const int NR = ceil_div(d->M, BM);
    const int NC = d->B * (d->ldt / BN);
    const int grid = 8 * NR * ceil_div(NC, 8);
    // fast path: direct-to-LDS high-occupancy kernel; the register-staged kernel remains for shapes outside its limits
    static const bool force_staged = getenv("SEPK_FORCE_STAGED") != nullptr;
    const bool direct_ok = !force_staged && d->K % DK == 0 && d->k_split % DK == 0 && d->M >= 4 && d->M % 4 == 0 &&
                           (d->pro_mode < SEP_PRO_GLN || d->K <= OMAXK);
    SEP_REQUIRE(direct_ok || (d->K % BK == 0 && d->k_split % BK == 0), "sep_pw_gemm: the register-staged fallback (K=%d) needs K %% 32 == 0", d->K);
    if (direct_ok) {
        // curated combinations exist in both arithmetics (GLN_BWD: fp32 MFMA only -- its time is the prologue, measured equal)
        const bool split3 = d->arith == SEP_ARITH_F16X3 && d->a_amax != nullptr;
        const bool split6 = d->arith == SEP_ARITH_BF16X6 || d->arith == SEP_ARITH_F16X3;
#define SEP_LD(T, P, S, E)                                                                                                                           \
    do {                                                                                                                                             \
        if (split3 && P != SEP_PRO_GLN_BWD)                                                                                                          \
            hipLaunchKernelGGL((pw_gemm_direct_kernel<T, P, S, E, (P != SEP_PRO_GLN_BWD ? 2 : 0)>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d); \
        else if (split6 && P != SEP_PRO_GLN_BWD)                                                                                                     \
            hipLaunchKernelGGL((pw_gemm_direct_kernel<T, P, S, E, (P != SEP_PRO_GLN_BWD ? 1 : 0)>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d); \
        else hipLaunchKernelGGL((pw_gemm_direct_kernel<T, P, S, E, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);                          \
    } while (0)
        // 1. the (operand form, prologue, split, epilogue) combinations of the Conv-TasNet step, epilogue flags compile-time
        const int ef = d->epi_flags, pm = d->pro_mode;
        const bool tr = d->trans_a != 0, sp = d->k_split != 0;
        bool done = true;
        if (d->M % BM != 0) done = false;          // the specialised epilogues have no row predicate
        else if (!tr && !sp && pm == SEP_PRO_NONE && ef == SEP_EPI_STATS_PRELU) SEP_LD(false, SEP_PRO_NONE, false, SEP_EPI_STATS_PRELU);              // TCN conv1
        else if (!tr && !sp && pm == SEP_PRO_GLN_PRELU && ef == SEP_EPI_RESIDUAL) SEP_LD(false, SEP_PRO_GLN_PRELU, false, SEP_EPI_RESIDUAL);     // heads
        else if (!tr && !sp && pm == SEP_PRO_GLN_PRELU && ef == 0) SEP_LD(false, SEP_PRO_GLN_PRELU, false, 0);                                   // last layer: skip head only
        else if (!tr && !sp && pm == SEP_PRO_PRELU && ef == SEP_EPI_SIGMOID) SEP_LD(false, SEP_PRO_PRELU, false, SEP_EPI_SIGMOID);               // mask
        else if (!tr && !sp && pm == SEP_PRO_PRELU && ef == 0) SEP_LD(false, SEP_PRO_PRELU, false, 0);                                           // mask without its activation (softmax over channels, staged layers)
        else if (!tr && !sp && pm == SEP_PRO_GLN && ef == 0) SEP_LD(false, SEP_PRO_GLN, false, 0);                                               // bottleneck
        else if (!tr && !sp && pm == SEP_PRO_NONE && ef == 0) SEP_LD(false, SEP_PRO_NONE, false, 0);                                             // plain 1x1 conv
        else if (!tr && !sp && pm == SEP_PRO_NONE && ef == SEP_EPI_RESIDUAL) SEP_LD(false, SEP_PRO_NONE, false, SEP_EPI_RESIDUAL);               // heads of the staged (causal) layers: input already normalised
        else if (tr && sp && pm == SEP_PRO_NONE && ef == 0) SEP_LD(true, SEP_PRO_NONE, true, 0);                                                 // ... and their heads^T
        else if (tr && !sp && pm == SEP_PRO_NONE && ef == 0) SEP_LD(true, SEP_PRO_NONE, false, 0);                                               // plain input gradient
        else if (tr && !sp && pm == SEP_PRO_NONE && ef == SEP_EPI_PRELU_BWD) SEP_LD(true, SEP_PRO_NONE, false, SEP_EPI_PRELU_BWD);               // mask^T
        else if (tr && sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LD(true, SEP_PRO_NONE, true, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);   // heads^T
        else if (tr && !sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LD(true, SEP_PRO_NONE, false, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);
        else if (tr && !sp && pm == SEP_PRO_NONE && ef == SEP_EPI_ROWSUMS) SEP_LD(true, SEP_PRO_NONE, false, SEP_EPI_ROWSUMS);                   // bottleneck^T
        else if (tr && !sp && pm == SEP_PRO_GLN_BWD && ef == SEP_EPI_RESIDUAL) SEP_LD(true, SEP_PRO_GLN_BWD, false, SEP_EPI_RESIDUAL);           // conv1^T
        else if (tr && !sp && pm == SEP_PRO_GLN_BWD && ef == 0) SEP_LD(true, SEP_PRO_GLN_BWD, false, 0);
        else done = false;
        // 2. anything else: same kernels with the flags read at run time
        if (!done) {
#define SEP_LG(T, P, S) hipLaunchKernelGGL((pw_gemm_direct_kernel<T, P, S, -1, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d)
#define SEP_LAUNCH_DIRECT(T, P)               \
    do {                                      \
        if (sp) SEP_LG(T, P, true);           \
        else SEP_LG(T, P, false);             \
    } while (0)
            switch (pm * 2 + (tr ? 1 : 0)) {
                case 0: SEP_LAUNCH_DIRECT(false, SEP_PRO_NONE); break;
                case 1: SEP_LAUNCH_DIRECT(true, SEP_PRO_NONE); break;
                case 2: SEP_LAUNCH_DIRECT(false, SEP_PRO_PRELU); break;
                case 3: SEP_LAUNCH_DIRECT(true, SEP_PRO_PRELU); break;
                case 4: SEP_LAUNCH_DIRECT(false, SEP_PRO_GLN); break;
                case 5: SEP_LAUNCH_DIRECT(true, SEP_PRO_GLN); break;
                case 6: SEP_LAUNCH_DIRECT(false, SEP_PRO_GLN_PRELU); break;
                case 7: SEP_LAUNCH_DIRECT(true, SEP_PRO_GLN_PRELU); break;
                case 8: SEP_LG(false, SEP_PRO_GLN_BWD, false); break;
                default: SEP_LG(true, SEP_PRO_GLN_BWD, false); break;
            }
#undef SEP_LAUNCH_DIRECT
#undef SEP_LG
        }
#undef SEP_LD
    } else
        hipLaunchKernelGGL(pw_gemm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
    SEP_CHECK_LAUNCH("sep_pw_gemm");
    return 0;
}

extern "C" int sep_pw_wgrad(const sep_wgrad_desc* d, sep_stream_t stream) {
    SEP_REQUIRE(d != nullptr, "sep_pw_wgrad: null descriptor");
    SEP_REQUIRE(d->B > 0 && d->M > 0 && d->N > 0 && d->T > 0, "sep_pw_wgrad: empty problem");
    SEP_REQUIRE(d->ldt % 128 == 0 && d->ldt >= d->T, "sep_pw_wgrad: ldt=%d must be a multiple of 128 and >= T=%d", d->ldt, d->T);
    SEP_REQUIRE(d->nsplit > 0 && (long)d->nsplit <= (long)d->B * (d->ldt / WK), "sep_pw_wgrad: bad nsplit=%d", d->nsplit);
    SEP_REQUIRE(d->g_split % BM == 0 && d->g_split < d->M, "sep_pw_wgrad: bad g_split=%d", d->g_split);
    SEP_REQUIRE(d->G && d->X && d->partial, "sep_pw_wgrad: null operand");
    SEP_REQUIRE(!d->g_split || d->G2, "sep_pw_wgrad: g_split without G2");
    SEP_REQUIRE(!d->g_mul || (d->Gaux && d->g_div > 0 && !d->g_split), "sep_pw_wgrad: g_mul needs Gaux/g_div");
    SEP_REQUIRE(d->x_div > 0, "sep_pw_wgrad: x_div must be >= 1");
    SEP_REQUIRE(d->x_mode >= 0 && d->x_mode <= SEP_PRO_GLN_PRELU, "sep_pw_wgrad: bad x_mode %d", d->x_mode);
    if (d->x_mode == SEP_PRO_PRELU || d->x_mode == SEP_PRO_GLN_PRELU) SEP_REQUIRE(d->x_alpha, "sep_pw_wgrad: needs x_alpha");
    if (d->x_mode >= SEP_PRO_GLN) SEP_REQUIRE(d->x_stats && d->x_gamma && d->x_beta && d->count > 0, "sep_pw_wgrad: gLN prologue needs stats/gamma/beta/count");
    const int ntiles = ceil_div(d->M, BM) * ceil_div(d->N, BN);
    const int grid = 8 * ntiles * ceil_div(d->nsplit, 8);
    static const bool force_staged = getenv("SEPK_FORCE_STAGED") != nullptr;
    const bool direct_ok = !force_staged && !d->g_mul && (d->x_mode < SEP_PRO_GLN || d->B / d->x_div <= WMAXB) &&
                           (long)d->nsplit <= (long)d->B * (d->ldt / DK);
    SEP_REQUIRE(d->arith >= SEP_ARITH_F32 && d->arith <= SEP_ARITH_F16X3, "sep_pw_wgrad: bad arith %d", d->arith);
    if (direct_ok && d->arith == SEP_ARITH_F16X3 && sep_pw_wgrad_pc16(d, (hipStream_t)stream)) {   // scaled two-part fp16 split (wgrad_pc16.hip)
        SEP_CHECK_LAUNCH("sep_pw_wgrad (pc16)");
        return 0;
    }
    if (direct_ok && d->arith != SEP_ARITH_F32 && sep_pw_wgrad_pc(d, (hipStream_t)stream)) {      // producer / consumer form (wgrad_pc.hip)
        SEP_CHECK_LAUNCH("sep_pw_wgrad (pc)");
        return 0;
    }
    if (direct_ok && d->arith != SEP_ARITH_F32) {      // F16X3: the weight gradient stays on the bf16 split (both operands are activations)
        switch (d->x_mode) {
            case SEP_PRO_NONE: hipLaunchKernelGGL((pw_wgrad_split_kernel<SEP_PRO_NONE>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d); break;
            case SEP_PRO_PRELU: hipLaunchKernelGGL((pw_wgrad_split_kernel<SEP_PRO_PRELU>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d); break;
            case SEP_PRO_GLN: hipLaunchKernelGGL((pw_wgrad_split_kernel<SEP_PRO_GLN>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d); break;
            default: hipLaunchKernelGGL((pw_wgrad_split_kernel<SEP_PRO_GLN_PRELU>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d); break;
        }
    } else if (direct_ok) {
        switch (d->x_mode) {
            case SEP_PRO_NONE: hipLaunchKernelGGL((pw_wgrad_direct_kernel<SEP_PRO_NONE>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *d); break;
            case SEP_PRO_PRELU: hipLaunchKernelGGL((pw_wgrad_direct_kernel<SEP_PRO_PRELU>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *d); break;
            case SEP_PRO_GLN: hipLaunchKernelGGL((pw_wgrad_direct_kernel<SEP_PRO_GLN>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *d); break;
            default: hipLaunchKernelGGL((pw_wgrad_direct_kernel<SEP_PRO_GLN_PRELU>), dim3(grid), dim3(512), 0, (hipStream_t)stream, *d); break;
        }
    } else
        hipLaunchKernelGGL(pw_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
    SEP_CHECK_LAUNCH("sep_pw_wgrad");
    return 0;
}

extern "C" int sep_pw_wgrad_batch(const sep_wgrad_desc* ds, int n, sep_stream_t stream) {
    SEP_REQUIRE(ds != nullptr && n >= 1 && n <= 8, "sep_pw_wgrad_batch: 1..8 products per call (got %d)", n);
    bool same = true;
    for (int k = 1; k < n; ++k) {
        const sep_wgrad_desc &a = ds[0], &b = ds[k];
        same = same && a.B == b.B && a.M == b.M && a.N == b.N && a.T == b.T && a.ldt == b.ldt && a.g_split == b.g_split && a.g_mul == b.g_mul &&
               a.g_div == b.g_div && a.x_mode == b.x_mode && a.x_div == b.x_div && a.nsplit == b.nsplit && a.arith == b.arith && a.eps == b.eps &&
               a.count == b.count && a.Gaux == b.Gaux && a.x_alpha == b.x_alpha && a.x_stats == b.x_stats && a.x_gamma == b.x_gamma &&
               a.x_beta == b.x_beta && (a.G2 == nullptr) == (b.G2 == nullptr) && (a.partial_bias == nullptr) == (b.partial_bias == nullptr);
    }
    SEP_REQUIRE(same, "sep_pw_wgrad_batch: the products must agree in everything but G, G2, X, partial, partial_bias");
    // one grid for all of them where the producer / consumer fp16 kernel takes the shape (checked like sep_pw_wgrad does, on the first)
    const sep_wgrad_desc* d = &ds[0];
    const bool plain_ok = d->B > 0 && d->M > 0 && d->N > 0 && d->T > 0 && d->ldt % 128 == 0 && d->ldt >= d->T && d->nsplit > 0 &&
                          (long)d->nsplit <= (long)d->B * (d->ldt / DK) && !d->g_mul && d->x_div == 1 && d->arith == SEP_ARITH_F16X3 &&
                          (!d->g_split || (d->g_split % BM == 0 && d->g_split < d->M)) && d->x_mode >= 0 && d->x_mode <= SEP_PRO_GLN_PRELU;
    bool ptrs_ok = plain_ok;
    for (int k = 0; k < n && ptrs_ok; ++k)
        ptrs_ok = ds[k].G && ds[k].X && ds[k].partial && (!d->g_split || ds[k].G2);
    if (ptrs_ok && (d->x_mode == SEP_PRO_NONE || ((d->x_mode == SEP_PRO_PRELU || d->x_mode == SEP_PRO_GLN_PRELU ? d->x_alpha != nullptr : true) &&
                                                   (d->x_mode >= SEP_PRO_GLN ? (d->x_stats && d->x_gamma && d->x_beta && d->count > 0) : true))) &&
        n > 1 && getenv("SEPK_WGRAD_BATCH_OFF") == nullptr && sep_pw_wgrad_pc16_batch(ds, n, (hipStream_t)stream)) {
        SEP_CHECK_LAUNCH("sep_pw_wgrad_batch (pc16)");
        return 0;
    }
    for (int k = 0; k < n; ++k) {               // any other shape / arithmetic: one launch per product, same results
        const int rc = sep_pw_wgrad(&ds[k], stream);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" int sep_reduce_slabs(const sep_reduce_seg* segs, int nseg, sep_stream_t stream) {
    SEP_REQUIRE(segs && nseg >= 1 && nseg <= RMAXSEG, "sep_reduce_slabs: 1..64 segments per launch (got %d)", nseg);
    ReduceArgs a;
    int blocks = 0;
    double launch_bytes = 0.0;                       // a launch that moves more than this streams: one thread per output is the faster form there
    for (int i = 0; i < nseg; ++i) launch_bytes += 4.0 * (double)segs[i].n * (double)segs[i].nslab;
    const bool streams = launch_bytes > 64e6;
    for (int i = 0; i < nseg; ++i) {
        SEP_REQUIRE(segs[i].src && segs[i].dst && segs[i].n > 0 && segs[i].nslab > 0, "sep_reduce_slabs: bad segment %d", i);
        a.seg[i] = segs[i];
        a.blk_start[i] = blocks;
        a.phased[i] = !streams && segs[i].nslab >= 16;
        blocks += ceil_div(segs[i].n, a.phased[i] ? 64 : 256);
    }
    a.blk_start[nseg] = blocks;
    a.nseg = nseg;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SEP_CHECK_LAUNCH("sep_reduce_slabs");
    return 0;
}

extern "C" int sep_f64_to_f32(const double* src, float* dst, int n, int accumulate, sep_stream_t stream) {
    SEP_REQUIRE(src && dst && n > 0, "sep_f64_to_f32: bad arguments");
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n, accumulate);
    SEP_CHECK_LAUNCH("sep_f64_to_f32");
    return 0;
}
