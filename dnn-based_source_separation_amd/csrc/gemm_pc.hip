// The 1x1-convolution GEMM of the Conv-TasNet step, packed-weight form (SEP_ARITH_F16X3 with sep_gemm_desc.A_pk), as a
// PRODUCER / CONSUMER workgroup: 512 threads = four consumer waves that do nothing but MFMAs (and the epilogue) and four
// producer waves that do everything else.  Same arithmetic and data formats as pw_gemm_coop_kernel (gemm_coop.hip): the
// weights arrive split by sep_pack_weights, X is put through the prologue and split once per workgroup, per-column
// power-of-two scales travel with the operands.  Replaces nn.Conv1d(kernel_size=1) of reference src/models/tdcn.py:86,173,175
// and src/models/conv_tasnet.py:335,341 (forward and input-gradient products).
//
// Why the roles are split (s_memtime stamps of the cooperative kernel, tools/coop_prof.py, heads shape): a wave spent
// ~375 cycles waiting for its operand reads, ~900 issuing 12 MFMAs + the split, ~350 at its waits and the barrier and ~330
// issuing five LDS-DMA pieces per 16-deep chunk -- strictly one after the other, since a wave issues in order, with only two
// or three waves per SIMD to overlap with: the matrix pipe was ~35 % busy although no single resource was exhausted.  Here a
// SIMD hosts ONE consumer wave, whose instruction stream is [barrier, ds_read next operands, 24 MFMAs] with the reads in
// flight under the MFMAs (two operand register sets), and ONE producer wave that issues the DMA pieces, reads the raw X
// chunk, applies the prologue, chooses the column scales, splits and writes the operands -- on the VALU / LDS / VMEM ports
// while the consumer's MFMAs own the matrix pipe.
//
// Tile: consumer wave 64 rows x 128 columns (2 x 4 accumulators of 32 x 32); workgroup 256 x 128 (consumers stacked 4 x 1,
// for M % 256 == 0) or 128 x 256 (2 x 2).  One workgroup per CU (94 - 120 KiB of LDS, <= 256 VGPRs).  One barrier per chunk:
//   producer step j : ds_read raw X(j) -> prologue, scale, split -> ds_write operands(j) -> wait DMA group j -> B_j -> issue group j+NS-1
//   consumer step j : B_j -> ds_read operands(j) (A from the DMA ring, X from the split buffer) -> MFMAs of chunk j-1
// DMA group g = {A chunk g, raw X chunk g+1}.
#include "gemm_common.hpp"
#include <stdlib.h>
#include <type_traits>

#ifdef PC_PROF
__device__ long long g_pc_prof[4096][8];      // [block][stamp]: wall clock (100 MHz) + shader clock stamps of wave 0 / wave 4
#define PSTAMP(w, s) do { if (wid == (w) && lane == 0 && bid < 4096) g_pc_prof[bid][s] = wall_clock64(); } while (0)
__device__ long long g_pc_step[2][64][8];      // [role][step][stamp] shader clock of one sampled workgroup
#define SSTAMP(role, s) do { if (bid == 808 && (wid & 3) == 1 && lane == 0 && j < 64) g_pc_step[role][j][s] = clock64(); } while (0)
extern "C" int sep_debug_pc_step(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_step), sizeof(long long) * 2 * 64 * 8) == hipSuccess ? 0 : -1; }
extern "C" int sep_debug_pc_prof(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_prof), sizeof(long long) * 4096 * 8) == hipSuccess ? 0 : -1; }
#else
#define PSTAMP(w, s) do { } while (0)
#define SSTAMP(role, s) do { } while (0)
#endif

#ifndef PC_SETPRIO
#define PC_SETPRIO 1
#endif

namespace {

constexpr int PCMAXK = 512;      // rows of the per-row affine table of the gLN prologues

template <int WR, int WC, int NS, bool AUX>
struct __attribute__((aligned(16))) PcSmem {
    static constexpr int TM = 64 * WR, TN = 128 * WC;
    static constexpr int XSTG = 16 * TN + 16;       // raw X stage: rows k >= 8 start 16 floats late (the two row halves of a
                                                    // column land 16 banks apart: conflict-free quad reads)
    double red[8];
    float Xr[NS][XSTG];                             // raw X ring as DMA'd
    float Cr[AUX ? NS : 1][AUX ? XSTG : 4];         // GLN_BWD: the pre-activation chunk
    float Bp[2][TN * 16];                           // split X chunk [col][4 x 16 B], same granule swizzle
    int be[2][TN];                                  // its per-column scale exponents
    float sc[PCMAXK];
    float sh[AUX ? 4 : PCMAXK];
};

__device__ __forceinline__ f32x16 pc_mfma(const u32x4_t a, const u32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void pc_wait_barrier() {      // vmcnt(N) lgkmcnt(0), then the workgroup barrier
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pc_lgkm0_barrier() {     // lgkmcnt(0) only (the consumer waves have no vector memory in flight)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int WR, int WC, int PRO, bool SPLIT, int EF, int NS>
__global__ __launch_bounds__(512, 2) void pw_gemm_pc_kernel(const sep_gemm_desc d) {
    constexpr bool P_PRELU = PRO == SEP_PRO_PRELU || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_GLN = PRO == SEP_PRO_GLN || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_BWD = PRO == SEP_PRO_GLN_BWD;
    using Smem = PcSmem<WR, WC, NS, P_BWD>;
    constexpr int TM = Smem::TM, TN = Smem::TN;
    constexpr int PX = 2 * WC;                            // X pieces per producer wave and chunk (TN / 16 / 4)
    constexpr int G = PX * (P_BWD ? 2 : 1);               // DMA instructions per producer wave and chunk
    // the GLN_BWD store-back shares vmcnt with the DMAs and may retire out of order with them: plain vmcnt(0) there
    constexpr int KEEP = P_BWD ? 0 : (NS - 2) * G;
    static_assert(KEEP < 64, "vmcnt field");
    __shared__ Smem sm;
    static_assert(sizeof(Smem) <= 160 * 1024, "LDS");
    static_assert(sizeof(Smem) - 64 >= 4 * EPI_WAVE_FLOATS * sizeof(float), "epilogue transpose buffer");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const bool prio_consumers = d.accumulate >= 0 && PC_SETPRIO;
    const int lk = lane >> 5, l31 = lane & 31;

    const int NR = d.M / TM;
    const int ntile_t = d.ldt / TN;
    const int NC = d.B * ntile_t;
    // XCD-aware decode: all row tiles of one column tile land on the same XCD (blockIdx % 8) and share X through its L2
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int rt = jj % NR;
    const int ct = (jj / NR) * 8 + xcd;
    if (ct >= NC) return;
    const int b = ct / ntile_t;
    const int t0 = (ct % ntile_t) * TN;
    const int m0 = rt * TM;
    const int nk = d.K / DK;
    PSTAMP(0, 0);

    // per-row affine of the prologue, once per workgroup (all eight waves fill it)
    float alpha_p = 0.f, mu = 0.f, rstd = 1.f, mg = 0.f, mgx = 0.f;
    if (P_PRELU || P_BWD) alpha_p = d.pro_alpha[0];
    if (P_GLN || P_BWD) {
        gln_mu_rstd(d.pro_stats + (size_t)b * SEP_STATS_SLOTS * 2, d.count, d.eps, mu, rstd);
        for (int k = tid; k < d.K; k += 512) {
            if (P_BWD) sm.sc[k] = d.pro_gamma[k];
            else {
                const float scv = d.pro_gamma[k] * rstd;
                sm.sc[k] = scv;
                sm.sh[k] = d.pro_beta[k] - mu * scv;
            }
        }
    }
    if (P_BWD) { mg = d.pro_bsum[2 * b]; mgx = d.pro_bsum[2 * b + 1]; }
    float dalpha_pro = 0.f;
    asm volatile("" :: "v"(alpha_p), "v"(mu), "v"(rstd), "v"(mg), "v"(mgx));      // loads consumed before the first asm DMA
    __syncthreads();                                     // tables visible; no DMA in flight yet

    f32x16 acc[2][2][2];                                 // [column half][mi][ni]: the consumer's 64 x 128 tile (producers: unused)

    if (producer) {
        // =================================================================================== producer waves
        const int pw = wid - 4;
        const int Ks1 = SPLIT ? d.k_split : d.K;
        const int split_chunk = SPLIT ? d.k_split / DK : -1;
        const size_t stepX = (size_t)DK * d.ldt;
        // X piece x = pw + 4q: TN = 128: contraction rows 2x, 2x+1 (two 512-byte rows); TN = 256: row x (one 1 KiB row)
        const unsigned offX = WC == 1 ? 4u * (unsigned)((lane >> 5) * d.ldt + 4 * (lane & 31)) : 16u * (unsigned)lane;
        const float* baseX = d.X + (size_t)b * Ks1 * d.ldt + t0;
        const float* baseC = P_BWD ? d.pro_aux + (size_t)b * Ks1 * d.ldt + t0 : nullptr;
        int xi = 0, xst = 0;                       // next chunk to issue and its ring stage
        auto issue_x = [&]() {
            if (SPLIT && xi == split_chunk) baseX = d.X2 + (size_t)b * (d.K - d.k_split) * d.ldt + t0;
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const int x = pw + 4 * q;
                const int krow = WC == 1 ? 2 * x : x;                                   // first contraction row of the piece
                const int ldo = (krow >= 8 ? 16 : 0) + krow * TN;
                glds16_asm(baseX + (size_t)krow * d.ldt, offX, lds_addr(&sm.Xr[xst][ldo]));
                if (P_BWD) glds16_asm(baseC + (size_t)krow * d.ldt, offX, lds_addr(&sm.Cr[P_BWD ? xst : 0][ldo]));
            }
            baseX += stepX;
            if (P_BWD) baseC += stepX;
            ++xi;
            xst = xst + 1 == NS ? 0 : xst + 1;
        };
        auto issue_group = [&]() {
            if (xi < nk) issue_x();
        };

        // this thread prepares contraction rows 8*kh .. 8*kh+7 of columns colb (+128): exactly one MFMA operand group each
        const int ptid = tid - 256;
        const int kh = ptid & 1, colb = ptid >> 1;
        const int r_off = (kh ? 16 : 0) + 8 * kh * TN + colb;                    // + j*TN (+128 for the second column)
        const int s_fsw = (colb >> 2) & 3;                                       // 128 is a multiple of 16: same swizzle for both columns
        const int w_hi = colb * 16 + 4 * ((2 * kh) ^ s_fsw);
        const int w_lo = colb * 16 + 4 * ((2 * kh + 1) ^ s_fsw);
        int bexp[WC];                                                            // scale exponent of the column ...
        bool bset[WC];                                                           // ... chosen yet?  (stays unset while the column has only seen zeros)
#pragma unroll
        for (int cc = 0; cc < WC; ++cc) { bexp[cc] = 0; bset[cc] = false; }
        const unsigned st_lane_off = 4u * (unsigned)(8 * kh * d.ldt + colb);     // GLN_BWD store-back: byte offset inside a chunk

        issue_x();
#pragma unroll
        for (int g = 0; g < NS - 1; ++g) issue_group();
        pc_wait_barrier<0>();                                                    // B_-1: raw X(0) (and groups 0 .. NS-2) landed
        PSTAMP(4, 1);

        int xstage = 0;
        for (int j = 0; j < nk; ++j) {
            SSTAMP(1, 0);
            const float* Xb = sm.Xr[xstage];
            const float* Cb = sm.Cr[P_BWD ? xstage : 0];
            float* Bp = sm.Bp[j & 1];
            float scv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, shv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (P_GLN || P_BWD) {
                const float4 s0 = ld4(&sm.sc[j * DK + 8 * kh]), s1 = ld4(&sm.sc[j * DK + 8 * kh + 4]);
                scv[0] = s0.x; scv[1] = s0.y; scv[2] = s0.z; scv[3] = s0.w; scv[4] = s1.x; scv[5] = s1.y; scv[6] = s1.z; scv[7] = s1.w;
            }
            if (P_GLN) {
                const float4 s0 = ld4(&sm.sh[j * DK + 8 * kh]), s1 = ld4(&sm.sh[j * DK + 8 * kh + 4]);
                shv[0] = s0.x; shv[1] = s0.y; shv[2] = s0.z; shv[3] = s0.w; shv[4] = s1.x; shv[5] = s1.y; shv[6] = s1.z; shv[7] = s1.w;
            }
#pragma unroll
            for (int cc = 0; cc < WC; ++cc) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = Xb[r_off + e * TN + 128 * cc];
                const bool live = t0 + colb + 128 * cc < d.T;
                if (P_BWD) {
                    float a8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) a8[e] = Cb[r_off + e * TN + 128 * cc];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // d(pre-activation) = rstd*(gamma_k*dv - mg - xhat*mgx) * PReLU'(a)
                        const float gk = scv[e];
                        const float a = a8[e];
                        const float u = prelu_f(a, alpha_p);
                        const float xh = (u - mu) * rstd;
                        const float du = rstd * (gk * v[e] - mg - xh * mgx);
                        const float da = live ? du * prelu_grad(a, alpha_p) : 0.f;
                        if (rt == 0) {
                            if (live && a <= 0.f) dalpha_pro += du * a;
                            float* srow = d.pro_store + ((size_t)b * d.K + j * DK + e) * d.ldt + t0 + 128 * cc;   // uniform row pointer + lane offset
                            *reinterpret_cast<float*>(reinterpret_cast<char*>(srow) + (size_t)st_lane_off) = da;
                        }
                        v[e] = da;
                    }
                } else if (PRO != SEP_PRO_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = v[e];
                        if (P_PRELU) x = prelu_f(x, alpha_p);
                        if (P_GLN) x = x * scv[e] + shv[e];
                        v[e] = x;
                    }
                }
                float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                                fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
                // the column's other 8 contraction rows sit in the neighbouring lane (quad_perm [1,0,3,2])
                m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
                const int ex = __builtin_amdgcn_frexp_expf(m);                   // m = f * 2^ex, f in [0.5, 1)
                if (m > 0.f && (!bset[cc] || ex + bexp[cc] > 14)) bexp[cc] = 9 - ex;      // first non-zero chunk, or the column outgrew its scale
                bset[cc] = bset[cc] || m > 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_ldexpf(v[e], bexp[cc]);
                unsigned hi[4], lo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) split2_pair(v[2 * q], v[2 * q + 1], hi[q], lo[q]);
                *reinterpret_cast<u32x4_t*>(Bp + w_hi + 128 * 16 * cc) = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4_t*>(Bp + w_lo + 128 * 16 * cc) = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
                if (kh == 0) sm.be[j & 1][colb + 128 * cc] = bexp[cc];
            }
            SSTAMP(1, 1);
#ifdef PC_PROF
            if (KEEP > 0 && j + NS <= nk) __builtin_amdgcn_s_waitcnt(0x0070 | (KEEP & 15) | ((KEEP >> 4) << 14)); else __builtin_amdgcn_s_waitcnt(0x0070);
            SSTAMP(1, 2);
#endif
            // raw X(j+1) has landed -- mine: all but the newer chunks; everyone's: the barrier -- and the operands of chunk j
            // are written
            if (KEEP > 0 && j + NS <= nk) pc_wait_barrier<KEEP>();
            else pc_wait_barrier<0>();                                           // B_j
            SSTAMP(1, 3);
            issue_group();                                                       // group j+NS-1 into the stages B_j freed
            SSTAMP(1, 4);
            xstage = xstage + 1 == NS ? 0 : xstage + 1;
        }
    } else {
        // =================================================================================== consumer waves
        const int wr = wid / WC, wcc = wid % WC;
        const int fsw = (l31 >> 2) & 3;
        const int c_hi = l31 * 16 + 4 * ((2 * lk) ^ fsw);                        // float offset inside a 32-column block image
        const int c_lo = l31 * 16 + 4 * ((2 * lk + 1) ^ fsw);
        const int b_base = 128 * wcc * 16;
        // A fragments come straight from the packed matrix (operand-block layout: 1 KiB per 32-row block, chunk and part,
        // lane-linear), one chunk ahead of their use: L2-resident weights, perfectly coalesced, no LDS and no barrier involved
        const char* Apk = reinterpret_cast<const char*>(d.A_pk) + (size_t)((m0 + 64 * wr) >> 5) * nk * 2048;     // wave-uniform
        const unsigned a_lane = 16u * (unsigned)lane;
        if (prio_consumers) __builtin_amdgcn_s_setprio(2);        // the MFMA stream wins the issue arbitration against its SIMD's producer wave
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][mi][n][r] = 0.f;
        // Operand registers: the A fragments of two consecutive chunks (SA[j & 1]) and TWO half-chunk B buffers (column blocks
        // 0-1 and 2-3).  Reads and MFMAs are staggered by half a chunk, so that every ds_read has 12 MFMAs (>= 384 cycles) of
        // cover and the consumer needs 64 operand registers beside its 128 accumulators (a full second operand set spilled).
        u32x4_t sa[2][2][2], sb[2][2][2];                                        // sa[chunk parity][mi][hi, lo]; sb[half][ni & 1][hi, lo]
        int en[2][2];                                                            // [half][ni & 1] scale exponents of the chunk's columns
        int bcur[4] = {0, 0, 0, 0};                                              // scale the accumulators of column block ni are in

        auto load_a = [&](auto parc, const int chunk) {
            constexpr int par = decltype(parc)::value;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const char* q = Apk + ((size_t)mi * nk + chunk) * 2048;                   // scalar arithmetic: the load takes the saddr form
                sa[par][mi][0] = *reinterpret_cast<const u32x4_t*>(q + a_lane);
                sa[par][mi][1] = *reinterpret_cast<const u32x4_t*>(q + 1024 + a_lane);
            }
        };
        auto read_b = [&](auto halfc, const int pb) {
            constexpr int h = decltype(halfc)::value;
            const float* Bb = sm.Bp[pb] + b_base + h * 64 * 16;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                sb[h][n][0] = *reinterpret_cast<const u32x4_t*>(Bb + n * 32 * 16 + c_hi);
                sb[h][n][1] = *reinterpret_cast<const u32x4_t*>(Bb + n * 32 * 16 + c_lo);
                en[h][n] = sm.be[pb][128 * wcc + 64 * h + 32 * n + l31];
            }
        };
        auto compute = [&](auto parc, auto halfc) {
            constexpr int par = decltype(parc)::value, h = decltype(halfc)::value;
            // the accumulators follow their column's scale
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int delta = en[h][n] - bcur[2 * h + n];
                if (__builtin_amdgcn_ballot_w64(delta != 0) != 0) {              // rare after the first chunks
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[h][mi][n][r] = __builtin_ldexpf(acc[h][mi][n][r], delta);
                }
                bcur[2 * h + n] = en[h][n];
            }
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const int asel = part == 1 ? 1 : 0, bsel = part == 0 ? 1 : 0;    // hi*lo, lo*hi, hi*hi
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[h][mi][n] = pc_mfma(sa[par][mi][asel], sb[h][n][bsel], acc[h][mi][n]);
            }
        };
        constexpr std::integral_constant<int, 0> I0{};
        constexpr std::integral_constant<int, 1> I1{};
        // chunk j after barrier B_j:  read B(j, half 0) | MFMAs (j-1, half 1) | load A(j+1) | read B(j, half 1) | MFMAs (j, half 0)
        // (A(j+1) goes into the registers chunk j-1 has just released and has ~24 MFMAs of cover before chunk j+1 needs it)
        load_a(I0, 0);
        pc_lgkm0_barrier();                                                      // B_-1
        int j = 0;
        for (; j + 1 < nk; j += 2) {
            SSTAMP(0, 0);
            pc_lgkm0_barrier();                                                  // B_j (j even)
            SSTAMP(0, 1);
            read_b(I0, 0);
            if (j > 0) compute(I1, I1);
#ifndef PC_ABL_NOA
            load_a(I1, j + 1);
#else
            if (j == 0) load_a(I1, 1);
#endif
            read_b(I1, 0);
            compute(I0, I0);
            SSTAMP(0, 2);
            pc_lgkm0_barrier();                                                  // B_j+1: my reads of chunk j are complete
            SSTAMP(0, 3);
            read_b(I0, 1);
            compute(I0, I1);
#ifndef PC_ABL_NOA
            if (j + 2 < nk) load_a(I0, j + 2);
#endif
            read_b(I1, 1);
            compute(I1, I0);
            SSTAMP(0, 4);
        }
        if (j < nk) {                                                            // odd chunk count: the last chunk has even parity
            pc_lgkm0_barrier();
            read_b(I0, 0);
            if (j > 0) compute(I1, I1);
            read_b(I1, 0);
            compute(I0, I0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            compute(I0, I1);
        } else {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            compute(I1, I1);
        }
        if (prio_consumers) __builtin_amdgcn_s_setprio(0);
        // undo the column scales (the row scales of A leave in the epilogue)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni >> 1][mi][ni & 1][r] = __builtin_ldexpf(acc[ni >> 1][mi][ni & 1][r], -bcur[ni]);
    }
    PSTAMP(0, 2);
    PSTAMP(4, 4);
    __syncthreads();                                    // ring reads done: the staging area becomes the transpose buffer
    // Launder what the epilogue derives its addresses from: hoisted above the main loop it would be held through it.
    int etid = tid, eb = b, em0 = m0, et0 = t0;
    asm volatile("" : "+v"(etid), "+s"(eb), "+s"(em0), "+s"(et0));
    const int ewid = __builtin_amdgcn_readfirstlane(etid >> 6);
    const int elane = etid & 63;
    const bool cons = ewid < 4;
    const int cw = cons ? ewid : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
        gemm_epilogue<EF, 2, true, 8, true>(d, acc[h], eb, em0, et0, cw / WC, 2 * (cw % WC) + h, elane >> 5, elane & 31, etid, &sm.Xr[0][0], sm.red, TN, cons);
#ifdef PC_PROF
    __builtin_amdgcn_s_waitcnt(0x0070);
    if (ewid == 0 && elane == 0 && blockIdx.x < 4096) g_pc_prof[blockIdx.x][3] = wall_clock64();
#endif
    if (P_BWD && rt == 0) {
        const double sdal = block_sum_n<double, 8>((double)dalpha_pro, sm.red);
        if (tid == 0) atomicAdd(d.pro_dalpha, sdal);
    }
}

template <int WR, int WC, int PRO, bool SPLIT, int EF>
void launch_pc(const sep_gemm_desc& d, const int ns, hipStream_t stream) {
    const int NR = d.M / (64 * WR);
    const int NC = d.B * (d.ldt / (128 * WC));
    const int grid = 8 * NR * ceil_div(NC, 8);
    constexpr bool BWD = PRO == SEP_PRO_GLN_BWD;
    if (ns == 5) hipLaunchKernelGGL((pw_gemm_pc_kernel<WR, WC, PRO, SPLIT, EF, (sizeof(PcSmem<WR, WC, 5, BWD>) <= 160 * 1024 ? 5 : 3)>), dim3(grid), dim3(512), 0, stream, d);
    else hipLaunchKernelGGL((pw_gemm_pc_kernel<WR, WC, PRO, SPLIT, EF, 3>), dim3(grid), dim3(512), 0, stream, d);
}

}  // namespace

// Called by sep_pw_gemm_packed (gemm_coop.hip).  Returns 1 when the call was launched here.
int sep_pw_gemm_pc(const sep_gemm_desc* d, hipStream_t stream) {
    static const int env_ns = getenv("SEPK_PC_NS") ? atoi(getenv("SEPK_PC_NS")) : 0;
    static const int force_22 = getenv("SEPK_PC_22") ? atoi(getenv("SEPK_PC_22")) : 0;
    if (d->M % 128 != 0 || d->K % DK != 0 || d->k_split % DK != 0 || (d->m_split % 128) != 0 || d->ldt % 256 != 0) return 0;
    if (d->pro_mode >= SEP_PRO_GLN && d->K > PCMAXK) return 0;
    if ((size_t)d->M * d->K * 4 >= (1ull << 32) || (size_t)4 * d->ldt * 4 >= (1ull << 31)) return 0;     // 32-bit DMA offsets
    const int ef = d->epi_flags, pm = d->pro_mode;
    const bool sp = d->k_split != 0;
    const bool tall = d->M % 256 == 0 && !force_22;      // 256 x 128 tile (consumers 4 x 1), else 128 x 256 (2 x 2)
    const int ns = env_ns == 3 || env_ns == 5 ? env_ns : (pm == SEP_PRO_GLN_BWD ? 5 : 3);      // measured: the deeper ring pays on the two-operand stream of the gLN-backward prologue only
#define SEP_LP(P, S, E)                                              \
    do {                                                             \
        if (tall) launch_pc<4, 1, P, S, E>(*d, ns, stream);          \
        else launch_pc<2, 2, P, S, E>(*d, ns, stream);               \
        return 1;                                                    \
    } while (0)
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_STATS_PRELU) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_STATS_PRELU);                      // TCN conv1
    if (!sp && pm == SEP_PRO_GLN_PRELU && ef == SEP_EPI_RESIDUAL) SEP_LP(SEP_PRO_GLN_PRELU, false, SEP_EPI_RESIDUAL);                 // heads
    if (!sp && pm == SEP_PRO_GLN_PRELU && ef == 0) SEP_LP(SEP_PRO_GLN_PRELU, false, 0);                                               // last layer: skip head only
    if (!sp && pm == SEP_PRO_PRELU && ef == SEP_EPI_SIGMOID) SEP_LP(SEP_PRO_PRELU, false, SEP_EPI_SIGMOID);                           // mask (sigmoid)
    if (!sp && pm == SEP_PRO_PRELU && ef == 0) SEP_LP(SEP_PRO_PRELU, false, 0);                                                       // mask (softmax follows)
    if (!sp && pm == SEP_PRO_GLN && ef == 0) SEP_LP(SEP_PRO_GLN, false, 0);                                                           // bottleneck
    if (!sp && pm == SEP_PRO_NONE && ef == 0) SEP_LP(SEP_PRO_NONE, false, 0);                                                         // plain 1x1 conv / input gradient
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_PRELU_BWD) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_PRELU_BWD);                         // mask^T
    if (sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LP(SEP_PRO_NONE, true, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);    // heads^T
    if (!sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);  // last layer's skip^T
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_ROWSUMS) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_ROWSUMS);                             // bottleneck^T
    if (!sp && pm == SEP_PRO_GLN_BWD && ef == SEP_EPI_RESIDUAL) SEP_LP(SEP_PRO_GLN_BWD, false, SEP_EPI_RESIDUAL);                     // conv1^T
    if (!sp && pm == SEP_PRO_GLN_BWD && ef == 0) SEP_LP(SEP_PRO_GLN_BWD, false, 0);
#undef SEP_LP
    return 0;
}
