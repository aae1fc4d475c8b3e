// The 1x1-convolution GEMM of the Conv-TasNet step in its packed-weight / cooperative-split form (SEP_ARITH_F16X3 with
// sep_gemm_desc.A_pk), and the weight packer that feeds it.
//
//   Y[b][m][t] = epi( a_rscale[m] * sum_k Apk[m][k] * 2^-e_t * split(2^e_t * pro(X[b][k][t])) + bias[m] )
//
// Replaces nn.Conv1d(kernel_size=1) of reference src/models/tdcn.py:86,173,175 and src/models/conv_tasnet.py:335,341
// (forward and input-gradient products) exactly like pw_gemm_direct_kernel<..., AR = 2> in gemm.hip, whose arithmetic
// (fp32 products from two fp16 parts per operand, three part products on v_mfma_f32_32x32x16_f16, fp32 accumulation,
// exact power-of-two scales) it keeps.  What changes is WHERE the operand splits are computed -- rocprofv3 on the round-1
// kernel showed the matrix pipe ~16 % busy and the VALU saturated by the splits (profiles/r01k_sq_counters.txt):
//
//  * A (the weights) is split ONCE per pass by sep_pack_weights into {hi[8], lo[8]} fp16 groups, 4 bytes per weight like
//    the fp32 matrix, with one scale per ROW of A (undone in the epilogue together with the bias add), in an operand-block
//    layout: the 1 KiB a wave needs for one 32-row block, chunk and part is contiguous and lane-linear, i.e. one
//    global_load_dwordx4 per lane IS the MFMA A operand -- no LDS copy, zero VALU.  The packer also writes the transposed /
//    concatenated forms, so the kernel has one operand path for forward and input gradients.
//  * X is put through the prologue (PReLU / gLN / gLN-backward) and split ONCE PER WORKGROUP: the four waves of a
//    workgroup are stacked along the rows (wave tile 32*MI x 64) over ONE 64-column tile, each thread prepares 4 of the
//    1024 values of a 16-deep chunk and writes them operand-ready to LDS, from where all four waves read them with
//    ds_read_b128.  Per MFMA the kernel issues ~1/6 of the VALU of the per-wave split.
//  * The per-column scale (a column of X is an accumulator column, owned by a lane) is chosen by the quad of threads that
//    prepares the column (two DPP max) and handed over with the operands; a consumer lane rescales its accumulators when
//    its column's exponent changes (rare after the first chunks).
//
// Pipeline, ONE barrier per 16-deep chunk s: [ds_read operands of s | ds_read raw X of s+1 | global_load A of s+1] -> the MFMAs
// of s with prologue, column scale, split and operand writes of s+1 woven between them (pinned with sched_barrier: an in-order
// wave that issues these phases one after the other leaves the matrix pipe idle through most of them) -> wait (raw X(s+2)
// landed) + barrier -> LDS-DMA of raw X(s+NS+1).  The raw X ring is the only LDS-DMA traffic: one instruction per wave and chunk.
#include "gemm_common.hpp"
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>


namespace {

#ifndef COOP_PIPE
#define COOP_PIPE true
#endif
constexpr int CBN = 64;          // columns (frames) per workgroup
constexpr int RBI = 264;         // floats per raw-X image of 4 contraction rows: 1 KiB + 32 B, so the four images of a chunk
                                 // start 8 banks apart and the quad-per-column reads below are conflict-free
constexpr int COMAXK = 512;      // rows of the per-row affine table of the gLN prologues

template <int MI, int NS, bool AUX>
struct __attribute__((aligned(16))) CoopSmem {
    double red[8];
    float Bs[NS][4 * RBI];                   // raw X chunk as DMA'd
    float Cs[AUX ? NS : 1][AUX ? 4 * RBI : 4];   // GLN_BWD: the pre-activation chunk
    float Bp[2][CBN * 16];                   // split X chunk [col][4 x 16 B], same granule swizzle
    int be[2][CBN];                          // its per-column scale exponents
    float sc[COMAXK];
    float sh[AUX ? 4 : COMAXK];
    float epi_pad[(4 * EPI_WAVE_FLOATS > NS * 4 * RBI + (AUX ? NS * 4 * RBI : 4) + 2 * CBN * 16 + 2 * CBN + COMAXK + (AUX ? 4 : COMAXK))
                      ? 4 * EPI_WAVE_FLOATS - (NS * 4 * RBI + (AUX ? NS * 4 * RBI : 4) + 2 * CBN * 16 + 2 * CBN + COMAXK + (AUX ? 4 : COMAXK)) : 4];
                                             // the epilogue transposes through this block from Bs on: keep it large enough
};

template <int MI, int NS, bool BWD>
constexpr int coop_occupancy() {
    // registers, not LDS, bound the residency now: 64 accumulators + two operand sets (64) + the splitter's values need ~160 VGPRs
    // with two 32-row blocks per wave (3 waves per SIMD), ~120 with one (4)
    constexpr int by_lds = (int)((160 * 1024) / ((sizeof(CoopSmem<MI, NS, BWD>) + 511) / 512 * 512));
    constexpr int by_regs = MI == 4 ? 2 : MI == 2 ? 3 : 4;
    return by_lds < by_regs ? by_lds : by_regs;
}

// single-instruction forms hipcc does not emit by itself (see gemm_pc.hip): v_max_f32 without the canonicalising v_max(x, x), the quad
// maximum as two dpp instructions, lo = x - float(hi half) as one mixed-precision FMA
__device__ __forceinline__ float co_vmax(const float a, const float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float co_quad_max(const float m) {               // max over the four lanes of a quad, in every lane
    float r, t;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(m));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(t));
    return r;
}
__device__ __forceinline__ void co_split2_pair(const float x0, const float x1, unsigned& hi, unsigned& lo) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(unsigned, h);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(x0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(x1), "v"(hi));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void pco_unroll(F&& f) {      // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); pco_unroll<N, I + 1>(f); }
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4_t a, const u32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// max over the four lanes of a quad, in every lane: two VALU with DPP operands
__device__ __forceinline__ float quad_max(float x) {
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
    return x;
}
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));      // vmcnt(N) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int MI, int PRO, bool SPLIT, int EF, int NS>
__global__ __launch_bounds__(256, (coop_occupancy<MI, NS, PRO == SEP_PRO_GLN_BWD>()))
void pw_gemm_coop_kernel(const sep_gemm_desc d) {
    constexpr bool P_PRELU = PRO == SEP_PRO_PRELU || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_GLN = PRO == SEP_PRO_GLN || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_BWD = PRO == SEP_PRO_GLN_BWD;
    constexpr int RW = 32 * MI;                          // rows per wave
    constexpr int BMc = 4 * RW;                          // rows per workgroup
    __shared__ CoopSmem<MI, NS, P_BWD> sm;
    using CoSmem = CoopSmem<MI, NS, P_BWD>;
    static_assert(sizeof(CoSmem) - offsetof(CoSmem, Bs) >= 4 * EPI_WAVE_FLOATS * sizeof(float), "epilogue transpose buffer");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 5, l31 = lane & 31;

    const int NR = d.M / BMc;
    const int ntile_t = d.ldt / CBN;
    const int NC = d.B * ntile_t;
    // XCD-aware decode: all row tiles of one column tile land on the same XCD (blockIdx % 8) and share X through its L2
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int rt = jj % NR;
    const int ct = (jj / NR) * 8 + xcd;
    if (ct >= NC) return;
    const int b = ct / ntile_t;
    const int t0 = (ct % ntile_t) * CBN;
    const int m0 = rt * BMc;
    const int nk = d.K / DK;
    const bool dead_tile = !P_BWD && t0 >= d.T;          // whole tile in the pad frames: outputs are zeros (block-uniform)

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
    // consume the loads NOW: the compiler does not count the asm LDS-DMAs below, so a wait it placed at a first use inside
    // the loop would be vmcnt(0) and drain the ring
    asm volatile("" :: "v"(alpha_p), "v"(mu), "v"(rstd), "v"(mg), "v"(mgx));

    // ---- X: LDS-DMA ring of raw chunks (wave-uniform base in an SGPR pair + a per-lane 32-bit byte offset).  A: straight from the
    // packed matrix into registers, one chunk ahead (operand-block layout: 1 KiB per 32-row block, chunk and part, lane-linear;
    // L2-resident, perfectly coalesced): no LDS copy of A, one DMA instruction per wave and chunk instead of five. ----------------
    const int Ks1 = SPLIT ? d.k_split : d.K;
    const int split_chunk = SPLIT ? d.k_split / DK : -1;
    const size_t stepX = (size_t)DK * d.ldt;
    const unsigned offX = 4u * (unsigned)((lane >> 4) * d.ldt + 4 * (lane & 15));
    const float* baseX = d.X + ((size_t)b * Ks1 + 4 * wid) * d.ldt + t0;
    const float* baseC = P_BWD ? d.pro_aux + ((size_t)b * Ks1 + 4 * wid) * d.ldt + t0 : nullptr;
    int xi = 0, xst = 0;                       // next chunk to issue and its ring stage
    auto issue_x = [&]() {
        if (SPLIT && xi == split_chunk) baseX = d.X2 + ((size_t)b * (d.K - d.k_split) + 4 * wid) * d.ldt + t0;
        glds16_asm_once(baseX, offX, lds_addr(&sm.Bs[xst][wid * RBI]));
        if (P_BWD) glds16_asm_once(baseC, offX, lds_addr(&sm.Cs[P_BWD ? xst : 0][wid * RBI]));
        baseX += stepX;
        if (P_BWD) baseC += stepX;
        ++xi;
        xst = xst + 1 == NS ? 0 : xst + 1;
    };
    const char* Apk = reinterpret_cast<const char*>(d.A_pk) + (size_t)((m0 + wid * RW) >> 5) * nk * 2048;     // wave-uniform
    const unsigned a_lane = 16u * (unsigned)lane;

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // ---- consumer side: this lane's operand granules (row / column l31 of a 32-block, contraction half lk) ------------
    const int fsw = (l31 >> 2) & 3;                                  // granule swizzle of row l31 (+ any multiple of 16)
    const int c_hi = l31 * 16 + 4 * ((2 * lk) ^ fsw);                // float offset inside a 32-row (32-column) block image
    const int c_lo = l31 * 16 + 4 * ((2 * lk + 1) ^ fsw);
    int bcur[2] = {0, 0};                                            // scale exponent the accumulators of column block ni are in

    // ---- splitter side: quad (lane & 3 = kq) of column col_s = 16*wid + (lane >> 2); this thread owns k = 4*kq .. 4*kq+3 --
    const int kq = lane & 3;
    const int col_s = 16 * wid + (lane >> 2);
    const int s_raw = kq * RBI + col_s;                              // + j*64
    const int s_fsw = (col_s >> 2) & 3;
    const int s_hi = col_s * 16 + 4 * ((2 * (kq >> 1)) ^ s_fsw) + 2 * (kq & 1);
    const int s_lo = col_s * 16 + 4 * ((2 * (kq >> 1) + 1) ^ s_fsw) + 2 * (kq & 1);
    const bool s_live = t0 + col_s < d.T;
    const unsigned st_lane_off = 4u * (unsigned)(4 * kq * d.ldt + col_s);   // GLN_BWD store-back: this thread's byte offset in a chunk
    constexpr int UNSET = 10000;                                     // scale exponent of a column that has only seen zeros: any first maximum
    int bexp = UNSET;                                                // "outgrows" it, and ldexp(0, UNSET) stays 0
    const bool fast_prelu = alpha_p >= 0.f && alpha_p <= 1.f;        // PReLU(x) = max(x, alpha x) there

    float raw[4], aux[4], v[4];
    float4 sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;
    auto read_raw = [&](const int stage, const int kn) {
        const float* Bb = sm.Bs[stage];
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = Bb[s_raw + j * 64];
        if (P_BWD) {
            const float* Cb = sm.Cs[P_BWD ? stage : 0];
#pragma unroll
            for (int j = 0; j < 4; ++j) aux[j] = Cb[s_raw + j * 64];
        }
        if (P_GLN || P_BWD) sc4 = ld4(&sm.sc[kn * DK + 4 * kq]);
        if (P_GLN) sh4 = ld4(&sm.sh[kn * DK + 4 * kq]);
    };
    // prologue of this thread's four values of chunk kn -> v[]
    auto pro_vals = [&](const int kn) {
        const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
        const float shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = raw[j];
            if (P_BWD) {
                // d(pre-activation) = rstd*(gamma_k*dv - mg - xhat*mgx) * PReLU'(a)
                const float a = aux[j];
                const float u = prelu_f(a, alpha_p);
                const float xh = (u - mu) * rstd;
                const float du = rstd * (scv[j] * x - mg - xh * mgx);
                const float da = s_live ? du * prelu_grad(a, alpha_p) : 0.f;
                if (rt == 0) {
                    if (s_live && a <= 0.f) dalpha_pro += du * a;
                    float* srow = d.pro_store + ((size_t)b * d.K + kn * DK + j) * d.ldt + t0;     // uniform row pointer + lane offset
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(srow) + (size_t)st_lane_off) = da;
                }
                x = da;
            } else {
                if (P_PRELU) x = fast_prelu ? co_vmax(x, alpha_p * x) : prelu_f(x, alpha_p);
                if (P_GLN) x = fmaf(x, scv[j], shv[j]);
            }
            v[j] = x;
        }
    };
    // column scale of the chunk (the column's 16 values sit in the four lanes of a quad): branch-free, see gemm_pc.hip
    auto scale_vals = [&]() {
        float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        m = co_quad_max(m);
        const int e = __builtin_amdgcn_frexp_expf(m);                    // m = f * 2^e, f in [0.5, 1)
        const int e2 = __builtin_bit_cast(int, m) == 0 ? -3 * UNSET : e; // a chunk of zeros never moves the scale
        bexp = e2 + bexp > 14 ? 9 - e : bexp;                            // first non-zero chunk, or the column outgrew its scale
    };
    unsigned h01, l01, h23, l23;
    auto split_vals = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_ldexpf(v[j], bexp);
        co_split2_pair(v[0], v[1], h01, l01);
        co_split2_pair(v[2], v[3], h23, l23);
    };
    auto write_vals = [&](const int pb) {
        float* Bp = sm.Bp[pb];
        *reinterpret_cast<uint2*>(Bp + s_hi) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(Bp + s_lo) = make_uint2(l01, l23);
        sm.be[pb][col_s] = bexp;                                         // the four lanes of the quad store the same word
    };

    // operand registers: A of two consecutive chunks (pa[s & 1]: loaded at the start of step s-1, a whole step ahead), B of the chunk
    // being multiplied (read at the start of its step; the other two or three waves of the SIMD cover that LDS round trip)
    u32x4_t pa[2][MI][2], pbv[2][2];
    int en[2];
    auto load_a = [&](auto setc, const int chunk) {
        constexpr int q = decltype(setc)::value;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const char* src = Apk + ((size_t)mi * nk + chunk) * 2048;
            pa[q][mi][0] = *reinterpret_cast<const u32x4_t*>(src + a_lane);
            pa[q][mi][1] = *reinterpret_cast<const u32x4_t*>(src + 1024 + a_lane);
        }
    };
    auto read_b = [&](const int pb) {
        const float* Bp = sm.Bp[pb];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            pbv[ni][0] = *reinterpret_cast<const u32x4_t*>(Bp + ni * 32 * 16 + c_hi);
            pbv[ni][1] = *reinterpret_cast<const u32x4_t*>(Bp + ni * 32 * 16 + c_lo);
            en[ni] = sm.be[pb][ni * 32 + l31];
        }
    };
    auto rescale = [&]() {                                               // the accumulators follow their column's scale
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int delta = en[ni] - bcur[ni];
            if (__builtin_amdgcn_ballot_w64(delta != 0) != 0) {          // rare after the first chunks
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], delta);
            }
            bcur[ni] = en[ni];
        }
    };
    constexpr int NM = 6 * MI;                                           // MFMAs per chunk: parts hi*lo, lo*hi, hi*hi over MI x 2 blocks
    auto M = [&](auto setc, auto ic) {
        constexpr int q = decltype(setc)::value, i = decltype(ic)::value;
        constexpr int part = i / (2 * MI), mi = (i % (2 * MI)) / 2, ni = i % 2;
        constexpr int asel = part == 1 ? 1 : 0, bsel = part == 0 ? 1 : 0;
        acc[mi][ni] = mfma_f16(pa[q][mi][asel], pbv[ni][bsel], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
    };
#define CO_I(k) std::integral_constant<int, (k)>{}
#define CO_SB() __builtin_amdgcn_sched_barrier(0)
    constexpr int GX = P_BWD ? 2 : 1;                                    // DMA instructions per wave and chunk
    // at the barrier of step s raw X(s+2) must have landed; younger in the queue and allowed to fly on: the A loads of chunk s+1 (issued
    // at the start of step s) and, with a ring of three, one more X group.  The GLN_BWD store-back shares vmcnt and may retire out of
    // order with the loads: plain vmcnt(0) there.
    constexpr int KEEP = P_BWD ? 0 : (NS - 2) * GX + 2 * MI;
    static_assert(KEEP < 64, "vmcnt field");
    // One pipeline step = one chunk: its MFMAs with everything else of the step woven between them and pinned -- an in-order wave that
    // issues [reads | split | 12 MFMAs | DMA] one after the other leaves the matrix pipe idle through three of the four phases
    // (s_memtime stamps of the first form: 1950 cycles per chunk for 384 cycles of MFMAs).
    auto step = [&](auto pc, const int s) {
        constexpr int P = decltype(pc)::value;
        const bool more = s + 1 < nk;
        read_b(s & 1);                                                   // operands of chunk s (written before the last barrier)
        if (more) {
            read_raw((s + 1) % NS, s + 1);
            load_a(CO_I(P ^ 1), s + 1);                                  // set P ^ 1 was multiplied in the previous step
        }
        CO_SB();
        rescale();
        CO_SB();
        pco_unroll<NM / 3>([&](auto ic) { M(CO_I(P), ic); });
        if (more) pro_vals(s + 1);
        CO_SB();
        pco_unroll<NM / 6>([&](auto ic) { M(CO_I(P), CO_I(decltype(ic)::value + NM / 3)); });
        if (more) scale_vals();
        CO_SB();
        pco_unroll<NM / 6>([&](auto ic) { M(CO_I(P), CO_I(decltype(ic)::value + NM / 2)); });
        if (more) split_vals();
        CO_SB();
        pco_unroll<NM / 6>([&](auto ic) { M(CO_I(P), CO_I(decltype(ic)::value + 2 * NM / 3)); });
        if (more) write_vals((s + 1) & 1);
        CO_SB();
        pco_unroll<NM / 6>([&](auto ic) { M(CO_I(P), CO_I(decltype(ic)::value + 5 * NM / 6)); });
        if (more) {
            // raw X(s+2) has landed -- mine: all but the younger loads; everyone's: the barrier --, the operands of chunk s+1 are
            // written, and every wave is past its reads of this step's stages
            if (KEEP > 0 && s + NS < nk) wait_vm_lgkm0_barrier<KEEP>();
            else wait_vm_lgkm0_barrier<0>();
            if (xi < nk) issue_x();                                      // X(s+NS+1) into the raw stage this step has read
        }
    };

    if (!dead_tile) {
        // ---- fill the raw ring: X0 .. X(NS-1); A(0) ----------------------------------------------------------------------
        __syncthreads();                               // prologue tables visible (no DMA in flight yet: drains nothing)
        issue_x();
#pragma unroll
        for (int g = 0; g < NS - 1; ++g)
            if (xi < nk) issue_x();
        load_a(CO_I(0), 0);
        wait_vm_lgkm0_barrier<0>();                    // X0 (and everything else) landed
        read_raw(0, 0);
        pro_vals(0);
        scale_vals();
        split_vals();
        write_vals(0);
        wait_vm_lgkm0_barrier<0>();                    // operands of chunk 0 visible; raw stage 0 free
        if (xi < nk) issue_x();                        // X(NS) -> raw stage 0
        int s = 0;
        for (; s + 1 < nk; s += 2) {
            step(CO_I(0), s);
            step(CO_I(1), s + 1);
        }
        if (s < nk) step(CO_I(0), s);
        // undo the column scales (the row scales of A leave in the epilogue)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = __builtin_ldexpf(acc[mi][ni][r], -bcur[ni]);
    }
    __syncthreads();                                    // ring reads done: the staging area becomes the transpose buffer
    // Launder what the epilogue derives its addresses from: hoisted above the main loop it would be held through it.
    int etid = tid, eb = b, em0 = m0, et0 = t0;
    asm volatile("" : "+v"(etid), "+s"(eb), "+s"(em0), "+s"(et0));
    const int ewid = __builtin_amdgcn_readfirstlane(etid >> 6);
    const int elane = etid & 63;
    gemm_epilogue<EF, MI, true, 4, COOP_PIPE>(d, acc, eb, em0, et0, ewid, 0, elane >> 5, elane & 31, etid, &sm.Bs[0][0], sm.red, CBN);
    if (P_BWD && rt == 0) {
        const double sdal = block_sum_256<double>((double)dalpha_pro, sm.red);
        if (tid == 0) atomicAdd(d.pro_dalpha, sdal);
    }
}

// ======================================================================================
// Weight packer: one wave per row of A.
// ======================================================================================
constexpr int PMAXSEG = 64;
struct PackArgs {
    sep_pack_seg seg[PMAXSEG];
    int blk_start[PMAXSEG + 1];
    int nseg;
};

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
    int sgi = 0;
    while (sgi + 1 < a.nseg && (int)blockIdx.x >= a.blk_start[sgi + 1]) ++sgi;
    const sep_pack_seg sg = a.seg[sgi];
    const int lane = threadIdx.x & 63;
    const int m = ((int)blockIdx.x - a.blk_start[sgi]) * 4 + (threadIdx.x >> 6);
    if (m >= sg.M) return;
    const int ng = sg.K / 8;                         // groups of 8 contraction indices; a lane owns groups lane, lane+64, ...
    float amax = 0.f;
    for (int g = lane; g < ng; g += 64) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * g + e;
            const float x = sg.trans ? sg.W[(size_t)k * sg.ldw + m] : sg.W[(size_t)m * sg.ldw + k];
            amax = fmaxf(amax, fabsf(x));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const int aexp = 13 - __builtin_amdgcn_frexp_expf(amax);      // A * 2^aexp < 2^13
    if (lane == 0) sg.rscale[m] = __builtin_ldexpf(1.f, -aexp);
    // operand-block layout: [m / 32][k / 16][hi | lo][lane = 32 * ((k >> 3) & 1) + (m & 31)][8 fp16] -- the 1 KiB a wave reads for
    // one 32-row block, 16-deep chunk and part is contiguous, lane-linear, and IS the MFMA A operand
    uint4* dst = reinterpret_cast<uint4*>(sg.dst) + (size_t)(m >> 5) * (ng >> 1) * 128 + (m & 31);
    for (int g = lane; g < ng; g += 64) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * g + e;
            x[e] = __builtin_ldexpf(sg.trans ? sg.W[(size_t)k * sg.ldw + m] : sg.W[(size_t)m * sg.ldw + k], aexp);
        }
        unsigned hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split2_pair(x[2 * q], x[2 * q + 1], hi[q], lo[q]);
        uint4* q = dst + (size_t)(g >> 1) * 128 + 32 * (g & 1);
        q[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        q[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

template <int MI, int PRO, bool SPLIT, int EF>
void launch_coop(const sep_gemm_desc& d, const int ns, hipStream_t stream) {
    const int NR = d.M / (128 * MI);
    const int NC = d.B * (d.ldt / CBN);
    const int grid = 8 * NR * ceil_div(NC, 8);
    if (ns == 3) hipLaunchKernelGGL((pw_gemm_coop_kernel<MI, PRO, SPLIT, EF, 3>), dim3(grid), dim3(256), 0, stream, d);
    else hipLaunchKernelGGL((pw_gemm_coop_kernel<MI, PRO, SPLIT, EF, 2>), dim3(grid), dim3(256), 0, stream, d);
}

}  // namespace

// Called by sep_pw_gemm (gemm.hip) for SEP_ARITH_F16X3 descriptors that carry packed weights.  Returns 1 when the call was
// launched here, 0 when the shape / flag combination is not one of the packed kernel's (the caller then uses A / A2).
#ifndef SEP_COOP_MI4_DEFAULT
#define SEP_COOP_MI4_DEFAULT 2      // heads^T on the 512 x 64 tile (every [dout; dS] value split once for all 512 outputs): -0.1 ms per step; conv1 gains nothing from it (r08g)
#endif
int sep_pw_gemm_packed(const sep_gemm_desc* d, hipStream_t stream) {
    static const bool off = getenv("SEPK_COOP") != nullptr && atoi(getenv("SEPK_COOP")) == 0;
    // SEPK_GEMM_KERNEL = auto (default): the producer / consumer kernel where it wins (long contractions), the cooperative
    // kernel elsewhere; "pc" / "coop": that kernel for every shape it takes (tests, A/B runs)
    static const char* kern = getenv("SEPK_GEMM_KERNEL");
    static const bool only_coop = kern != nullptr && kern[0] == 'c';
    static const bool all_pc = kern != nullptr && kern[0] == 'p';
    static const int pc_min_k = getenv("SEPK_PC_MINK") ? atoi(getenv("SEPK_PC_MINK")) : 512;
    static const int force_mi = getenv("SEPK_COOP_MI") ? atoi(getenv("SEPK_COOP_MI")) : 0;
    static const int env_ns = getenv("SEPK_COOP_NS") ? atoi(getenv("SEPK_COOP_NS")) : 0;
    if (off || !d->A_pk || !d->a_rscale || d->arith != SEP_ARITH_F16X3) return 0;
    // measured per shape (tools/gemm_bench.py --packed, SEPK_GEMM_KERNEL=pc vs the default): the one-workgroup-per-CU kernel wins
    // where the main loop is long (K >= 512) or the row tiles are many (M >= 1024); at K = 128 / 256 its un-overlapped ring
    // fill and epilogue per tile cost more than its denser MFMA stream gains
    if (!only_coop && (all_pc || d->K >= pc_min_k || d->M >= 1024) && sep_pw_gemm_pc(d, stream)) return 1;      // producer / consumer form (gemm_pc.hip)
    // everything else the packed path takes runs on this file's cooperative kernel: ~5 % behind the per-wave-split kernel of
    // gemm.hip on the K = 128 shapes, but with the packer's per-row weight scales instead of one bound for all weights
    if (d->M % 128 != 0 || d->K % DK != 0 || d->k_split % DK != 0 || (d->m_split % 128) != 0) return 0;
    if (d->pro_mode >= SEP_PRO_GLN && d->K > COMAXK) return 0;
    if ((size_t)d->M * d->K * 4 >= (1ull << 32) || (size_t)4 * d->ldt * 4 >= (1ull << 31)) return 0;     // 32-bit DMA offsets
    const int ef = d->epi_flags, pm = d->pro_mode;
    const bool sp = d->k_split != 0;
    const int mi = (force_mi == 1 || d->M % 256 != 0) ? 1 : 2;
    const int ns = env_ns == 2 || env_ns == 3 ? env_ns : 2;
    // 128-row waves (a 512 x 64 workgroup tile: every value of X is split ONCE for 512 outputs instead of once per 256): the write-heavy
    // short-contraction shapes with M % 512 == 0 -- TCN conv1, heads^T, skip^T (SEPK_COOP_MI=4 while it is being measured)
    // SEPK_COOP_MI4: bit 0 = conv1, bit 1 = heads^T / skip^T (two sources), bit 2 = the plain product; SEPK_COOP_MI=4 = all three.
    // Measured, recorded sequence (profiles/r08f_kernel_choice_toggles.txt, r08g_coop_tile_per_shape.txt): none 15.29, conv1 15.33, heads^T 15.19, all 15.22 ms per step
    // (means of three alternating runs on one box).
    static const int mi4 = force_mi == 4 ? 7 : getenv("SEPK_COOP_MI4") ? atoi(getenv("SEPK_COOP_MI4")) : SEP_COOP_MI4_DEFAULT;
    if ((mi4 & 1) && force_mi != 1 && force_mi != 2 && d->M % 512 == 0 && !sp && pm == SEP_PRO_NONE && ef == SEP_EPI_STATS_PRELU) { launch_coop<4, SEP_PRO_NONE, false, SEP_EPI_STATS_PRELU>(*d, ns, stream); return 1; }
    if ((mi4 & 2) && force_mi != 1 && force_mi != 2 && d->M % 512 == 0 && sp && pm == SEP_PRO_NONE && ef == 0) { launch_coop<4, SEP_PRO_NONE, true, 0>(*d, ns, stream); return 1; }
    if ((mi4 & 4) && force_mi != 1 && force_mi != 2 && d->M % 512 == 0 && !sp && pm == SEP_PRO_NONE && ef == 0) { launch_coop<4, SEP_PRO_NONE, false, 0>(*d, ns, stream); return 1; }
#define SEP_LC(P, S, E)                                              \
    do {                                                             \
        if (mi == 2) launch_coop<2, P, S, E>(*d, ns, stream);        \
        else launch_coop<1, P, S, E>(*d, ns, stream);                \
        return 1;                                                    \
    } while (0)
    // the (prologue, two-source contraction, epilogue) combinations of the Conv-TasNet step, epilogue flags compile-time
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_STATS_PRELU) SEP_LC(SEP_PRO_NONE, false, SEP_EPI_STATS_PRELU);                      // TCN conv1
    if (!sp && pm == SEP_PRO_GLN_PRELU && ef == SEP_EPI_RESIDUAL) SEP_LC(SEP_PRO_GLN_PRELU, false, SEP_EPI_RESIDUAL);                 // heads
    if (!sp && pm == SEP_PRO_GLN_PRELU && ef == 0) SEP_LC(SEP_PRO_GLN_PRELU, false, 0);                                               // last layer: skip head only
    if (!sp && pm == SEP_PRO_PRELU && ef == SEP_EPI_SIGMOID) SEP_LC(SEP_PRO_PRELU, false, SEP_EPI_SIGMOID);                           // mask (sigmoid)
    if (!sp && pm == SEP_PRO_PRELU && ef == 0) SEP_LC(SEP_PRO_PRELU, false, 0);                                                       // mask (softmax follows)
    if (!sp && pm == SEP_PRO_GLN && ef == 0) SEP_LC(SEP_PRO_GLN, false, 0);                                                           // bottleneck
    if (!sp && pm == SEP_PRO_NONE && ef == 0) SEP_LC(SEP_PRO_NONE, false, 0);                                                         // plain 1x1 conv / input gradient
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_PRELU_BWD) SEP_LC(SEP_PRO_NONE, false, SEP_EPI_PRELU_BWD);                         // mask^T
    if (sp && pm == SEP_PRO_NONE && ef == 0) SEP_LC(SEP_PRO_NONE, true, 0);                                                           // heads^T (its gLN sums come from the weight gradient)
    if (sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LC(SEP_PRO_NONE, true, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);    // heads^T
    if (!sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LC(SEP_PRO_NONE, false, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);  // last layer's skip^T
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_ROWSUMS) SEP_LC(SEP_PRO_NONE, false, SEP_EPI_ROWSUMS);                             // bottleneck^T
    if (!sp && pm == SEP_PRO_GLN_BWD && ef == SEP_EPI_RESIDUAL) SEP_LC(SEP_PRO_GLN_BWD, false, SEP_EPI_RESIDUAL);                     // conv1^T
    if (!sp && pm == SEP_PRO_GLN_BWD && ef == 0) SEP_LC(SEP_PRO_GLN_BWD, false, 0);
#undef SEP_LC
    return 0;
}

extern "C" int sep_pack_weights(const sep_pack_seg* segs, int nseg, sep_stream_t stream) {
    SEP_REQUIRE(segs && nseg >= 1, "sep_pack_weights: no segments");
    for (int s0 = 0; s0 < nseg; s0 += PMAXSEG) {
        PackArgs a;
        const int n = nseg - s0 < PMAXSEG ? nseg - s0 : PMAXSEG;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            const sep_pack_seg& s = segs[s0 + i];
            SEP_REQUIRE(s.W && s.dst && s.rscale && s.M > 0 && s.K > 0 && s.K % 16 == 0 && s.M % 32 == 0 && s.ldw > 0, "sep_pack_weights: bad segment %d (M=%d K=%d)", s0 + i, s.M, s.K);
            SEP_REQUIRE((reinterpret_cast<size_t>(s.dst) & 31) == 0, "sep_pack_weights: dst of segment %d is not 32-byte aligned", s0 + i);
            a.seg[i] = s;
            a.blk_start[i] = blocks;
            blocks += ceil_div(s.M, 4);
        }
        a.blk_start[n] = blocks;
        a.nseg = n;
        hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        SEP_CHECK_LAUNCH("sep_pack_weights");
    }
    return 0;
}
