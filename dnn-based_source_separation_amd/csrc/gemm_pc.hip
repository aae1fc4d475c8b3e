// The 1x1-convolution GEMM of the Conv-TasNet step, packed-weight form (SEP_ARITH_F16X3 with sep_gemm_desc.A_pk), as a
// PRODUCER / CONSUMER workgroup: 512 threads = four consumer waves (one per SIMD) that do nothing but MFMAs, operand reads and the
// epilogue on a 64 x 128 accumulator tile each, and four producer waves that fetch X, apply the prologue, choose the per-column
// power-of-two scales, split into {hi, lo} fp16 and write the MFMA operands.  Weights arrive split by sep_pack_weights
// (gemm_coop.hip) and go straight from global memory into the consumers' registers.  Workgroup tile 256 x 128 (consumers 4 x 1,
// M % 256 == 0) or 128 x 256 (2 x 2).  Replaces nn.Conv1d(kernel_size=1) of reference src/models/tdcn.py:86,173,175 and
// src/models/conv_tasnet.py:335,341 (forward and input-gradient products).
//
// There is NO workgroup barrier in the main loop.  The first form of this kernel (round 2, commit 9b662e0) met at one
// s_barrier per chunk; its disassembly shows the consumer's s_barrier sitting in the middle of its MFMA stream with the
// matrix pipe drained behind it (a wave cannot issue past a barrier, and an in-order wave has at most one MFMA in flight
// when it gets there), and every chunk costs max(producer, consumer) + the barrier's skew: stamps gave 1425 cycles per
// chunk for 768 cycles of MFMAs.  Here
//   * every producer wave owns 32 (x WC) frame columns for the whole contraction: it DMAs ITS columns' rows into a
//     wave-private raw ring (so the only wait is its own vmcnt), keeps the columns' scale state in registers as before,
//     and publishes the split operands of chunk j in slot j % NB of a ring of NB operand buffers by storing j + 1 to
//     ready[wave] -- a plain LDS store; the LDS executes one wave's instructions in order, so whoever reads j + 1 there
//     finds the operands written;
//   * a consumer wave samples the four `ready` words half a chunk before it needs them (one ds_read_b128 under 12 MFMAs),
//     spins only if the producers are late, and stores j + 1 to freed[wave] behind its last operand read of chunk j; a
//     producer looks at `freed` before it overwrites slot (j + NB) % NB.
// Producers run up to NB - 1 chunks ahead, so jitter on either side is absorbed instead of being paid at a barrier, and
// the consumer's instruction order [operand reads | 12 MFMAs | A loads, operand reads, signals | 12 MFMAs] is pinned with
// sched_barrier (left alone, hipcc sank the reads behind the MFMAs that were meant to cover them).
#include "gemm_common.hpp"
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

#ifdef PCD_PROF
__device__ long long g_pcd_prof[4096][16];
extern "C" int sep_debug_pcd_prof(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pcd_prof), sizeof(long long) * 4096 * 16) == hipSuccess ? 0 : -1; }
#define PCD_STAMP(w, s, val) do { if (wid == (w) && lane == 0 && bid < 4096) g_pcd_prof[bid][s] = (val); } while (0)
#define PCD_COUNT(x) (x)
#else
#define PCD_STAMP(w, s, val) do { } while (0)
#define PCD_COUNT(x) do { } while (0)
#endif


namespace {

constexpr int PCDMAXK = 512;      // rows of the per-row affine table of the gLN prologues

template <int WR, int WC, int NS, int NB, bool AUX>
struct __attribute__((aligned(16))) PcdSmem {
    static constexpr int TM = 64 * WR, TN = 128 * WC;
    static constexpr int PIECE = 8 * 32 + 32;       // one DMA instruction: 8 contraction rows x 32 columns, lane-linear; + 32 floats so that the two
                                                    // row halves of a column sit 32 banks apart (conflict-free reads by (column, half) lanes)
    static constexpr int WSTG = 2 * WC * PIECE;     // raw stage of ONE producer wave: [column block][row half]
    double red[8];
    int ready[4];                                   // ready[p]: chunks published by producer wave p
    int freed[4];                                   // freed[c]: chunks consumer wave c has finished reading
    float Xr[4][NS][WSTG];                          // raw X rings, one per producer wave
    float Cr[AUX ? 4 : 1][AUX ? NS : 1][AUX ? WSTG : 4];      // GLN_BWD: the pre-activation chunk, same layout
    float Bp[NB][TN * 16];                          // split X chunks [col][4 x 16 B] (granule XOR swizzle, see the producer's w_hi / w_lo)
    int be[NB][TN];                                 // their per-column scale exponents
    float sc[PCDMAXK];
    float sh[AUX ? 4 : PCDMAXK];
};
template <int WR, int WC, int NS, int NB, bool AUX>
constexpr bool pcd_fits() { return sizeof(PcdSmem<WR, WC, NS, NB, AUX>) <= 160 * 1024; }

__device__ __forceinline__ f32x16 pcd_mfma(const u32x4_t a, const u32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
typedef int pcd_i4 __attribute__((ext_vector_type(4)));
// smallest of four progress counters (one 16-byte LDS read, same address in every lane)
#define PCD_LDS __attribute__((address_space(3)))
__device__ __forceinline__ pcd_i4 pcd_load4(const int* c) {
    return *(const volatile PCD_LDS pcd_i4*)(const PCD_LDS int*)c;      // explicit LDS pointer: a generic volatile access becomes a FLAT one
}
__device__ __forceinline__ int pcd_min(const pcd_i4 v) { return __builtin_amdgcn_readfirstlane(min(min(v.x, v.y), min(v.z, v.w))); }
__device__ __forceinline__ int pcd_min4(const int* c) { return pcd_min(pcd_load4(c)); }
__device__ __forceinline__ void pcd_post(int* c, const int value) {
    *(volatile PCD_LDS int*)(PCD_LDS int*)c = value;
}
#define PCD_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// Single-instruction forms the compiler does not produce by itself (a producer wave is issue-bound, see the kernel):
// v_max_f32 without the canonicalising v_max(x, x) in front, the neighbour-lane maximum as ONE dpp instruction (the s_nop
// covers the VALU-write -> DPP-read hazard the compiler cannot see behind asm), and lo = x - float(hi half) as one
// mixed-precision FMA (exact: the fp16 operand is extended, the arithmetic is fp32).
__device__ __forceinline__ float pcd_vmax(const float a, const float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float pcd_max_quad_neighbour(const float m) {      // max(m, m of lane ^ 1)
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(m));
    return r;
}
__device__ __forceinline__ void pcd_split2_pair(const float x0, const float x1, unsigned& hi, unsigned& lo) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(unsigned, h);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(x0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(x1), "v"(hi));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
constexpr int pcd_gcd(int a, int b) { return b == 0 ? a : pcd_gcd(b, a % b); }
template <int N, int I = 0, class F>
__device__ __forceinline__ void pcd_unroll(F&& f) {      // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); pcd_unroll<N, I + 1>(f); }
}

template <int WR, int WC, int PRO, bool SPLIT, int EF, int NS, int NB>
__global__ __launch_bounds__(512, 2) void pw_gemm_pc_kernel(const sep_gemm_desc d) {
    constexpr bool P_PRELU = PRO == SEP_PRO_PRELU || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_GLN = PRO == SEP_PRO_GLN || PRO == SEP_PRO_GLN_PRELU;
    constexpr bool P_BWD = PRO == SEP_PRO_GLN_BWD;
    using Smem = PcdSmem<WR, WC, NS, NB, P_BWD>;
    constexpr int TM = Smem::TM, TN = Smem::TN, PIECE = Smem::PIECE;
    constexpr int G = 2 * WC * (P_BWD ? 2 : 1);           // DMA instructions per producer wave and chunk
    // the GLN_BWD store-back shares vmcnt with the DMAs and may retire out of order with them: plain vmcnt(0) there
    constexpr int KEEP = P_BWD ? 0 : (NS - 1) * G;
    static_assert(KEEP < 64, "vmcnt field");
    static_assert(NB >= 2 && NS >= 2, "ring depths");
    __shared__ Smem sm;
    static_assert(sizeof(Smem) <= 160 * 1024, "LDS");
    static_assert(sizeof(Smem) - offsetof(Smem, Xr) >= 4 * EPI_WAVE_FLOATS * sizeof(float), "epilogue transpose buffer");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
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
    PCD_STAMP(0, 0, wall_clock64());
    long long n_miss = 0, n_spin = 0, c_vm = 0, c_loop = 0;

    // per-row affine of the prologue, once per workgroup (all eight waves fill it)
    float alpha_p = 0.f, mu = 0.f, rstd = 1.f, mg = 0.f, mgx = 0.f;
    if (P_PRELU || P_BWD) alpha_p = d.pro_alpha[0];
    if (P_GLN || P_BWD) {
        gln_mu_rstd(d.pro_stats + (size_t)b * SEP_STATS_SLOTS * 2, d.count, d.eps, mu, rstd);
        for (int k = tid; k < d.K; k += 512) {
            if (P_BWD) sm.sc[k] = rstd * d.pro_gamma[k];
            else {
                const float scv = d.pro_gamma[k] * rstd;
                sm.sc[k] = scv;
                sm.sh[k] = d.pro_beta[k] - mu * scv;
            }
        }
    }
    // the two means of the gLN being back-propagated: published floats (pro_bsum), or formed here from the producer's slots (pro_bacc) by
    // ONE thread and handed over through LDS -- in every thread that code cost the main loop spilled registers (the kernel sits at 256 VGPRs)
    const bool from_slots = P_BWD && d.pro_bacc != nullptr;
    if (P_BWD && !from_slots) { mg = d.pro_bsum[2 * b]; mgx = d.pro_bsum[2 * b + 1]; }
    if (from_slots && tid == 64) {
        float a, c;
        gln_bwd_means(d.pro_bacc + (size_t)b * SEP_STATS_SLOTS * 2, d.pro_stats + (size_t)b * SEP_STATS_SLOTS * 2, d.count, d.eps, a, c);
        float* hand = reinterpret_cast<float*>(sm.red);
        hand[0] = a; hand[1] = c;
    }
    if (tid < 4) { sm.ready[tid] = 0; sm.freed[tid] = 0; }
    float dalpha_pro = 0.f;
    asm volatile("" :: "v"(alpha_p), "v"(mu), "v"(rstd), "v"(mg), "v"(mgx));      // loads consumed before the first asm DMA
    __syncthreads();                                     // tables, counters and the two means visible; no DMA in flight yet
    if (from_slots) {
        const float* hand = reinterpret_cast<const float*>(sm.red);
        mg = hand[0]; mgx = hand[1];
        asm volatile("" :: "v"(mg), "v"(mgx));
    }

    f32x16 acc[2][2][2];                                 // [column half][mi][ni]: the consumer's 64 x 128 tile (producers: unused)

    if (producer) {
        // =================================================================================== producer waves
        // A wave issues at most one instruction every ~4 cycles, and ONE producer wave per SIMD prepares a quarter of every
        // chunk: its instruction COUNT per chunk is what the main loop waits for (measured: 187 instructions -> 1280 cycles per
        // chunk against 768 cycles of MFMAs).  Hence: the loop is unrolled over lcm(NS, NB) chunks (ring stages, slots and
        // the register ping-pong become constants), PReLU has a two-instruction form for 0 <= alpha <= 1, the gLN-backward
        // prologue is three FMAs on pre-multiplied constants, and the scale logic is branch-free.
        const int pw = wid - 4;
        const int Ks1 = SPLIT ? d.k_split : d.K;
        const int split_chunk = SPLIT ? d.k_split / DK : -1;
        const size_t stepX = (size_t)DK * d.ldt;
        // one DMA instruction = contraction rows 8*hh .. 8*hh+7 of this wave's 32 columns of block cc: lane -> (row lane >> 3, 16 B lane & 7)
        const unsigned offX = 4u * (unsigned)((lane >> 3) * d.ldt + 4 * (lane & 7));
        const float* baseX = d.X + (size_t)b * Ks1 * d.ldt + t0 + 32 * pw;
        const float* baseC = P_BWD ? d.pro_aux + (size_t)b * Ks1 * d.ldt + t0 + 32 * pw : nullptr;
        const unsigned xr_lds = lds_addr(&sm.Xr[pw][0][0]);
        const unsigned cr_lds = lds_addr(&sm.Cr[P_BWD ? pw : 0][0][0]);
        int xi = 0;                                // next chunk to issue
        auto issue_x = [&](auto stc) {
            constexpr int st = decltype(stc)::value;
            if (SPLIT && xi == split_chunk) baseX = d.X2 + (size_t)b * (d.K - d.k_split) * d.ldt + t0 + 32 * pw;
#pragma unroll
            for (int cc = 0; cc < WC; ++cc)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const size_t go = (size_t)(8 * hh) * d.ldt + 128 * cc;
                    glds16_asm_once(baseX + go, offX, xr_lds + 4u * (unsigned)(st * Smem::WSTG + (2 * cc + hh) * PIECE));
                    if (P_BWD) glds16_asm_once(baseC + go, offX, cr_lds + 4u * (unsigned)(st * Smem::WSTG + (2 * cc + hh) * PIECE));
                }
            baseX += stepX;
            if (P_BWD) baseC += stepX;
            ++xi;
        };

        // this thread prepares contraction rows 8*kh .. 8*kh+7 of column colb (+128): exactly one MFMA operand group each
        const int kh = lane & 1, c = lane >> 1;
        const int colb = 32 * pw + c;
        const int r_off = kh * PIECE + c;                                        // + 2*PIECE*cc + 32*e
        const int s_fsw = (colb >> 2) & 3;                                       // 128 is a multiple of 16: same swizzle for both columns
        const int w_hi = colb * 16 + 4 * ((2 * kh) ^ s_fsw);
        const int w_lo = colb * 16 + 4 * ((2 * kh + 1) ^ s_fsw);
        constexpr int UNSET = 10000;                                             // scale exponent of a column that has only seen zeros: any first maximum
        int bexp[WC];                                                            // "outgrows" it (ex + UNSET > 14), and ldexp(0, UNSET) stays 0
#pragma unroll
        for (int cc = 0; cc < WC; ++cc) bexp[cc] = UNSET;
        const unsigned st_lane_off = 4u * (unsigned)(8 * kh * d.ldt + colb);     // GLN_BWD store-back: byte offset inside a chunk
        const bool edge = t0 + TN > d.T;                                         // only the last column tile of a sample has dead frames
        bool live[WC];
#pragma unroll
        for (int cc = 0; cc < WC; ++cc) live[cc] = t0 + colb + 128 * cc < d.T;
        // gLN backward: d(pre-activation) = rstd*(gamma_k*dv - mg - xhat*mgx) * PReLU'(a), xhat = (PReLU(a) - mu)*rstd
        //                                 = (dv * (rstd*gamma_k) + PReLU(a) * k1 + k0) * PReLU'(a);   sc[] holds rstd*gamma_k
        const float bk1 = -rstd * rstd * mgx, bk0 = rstd * (mu * rstd * mgx - mg);

        float v[2][WC][8], a8[2][WC][8];                                         // raw chunk j (and j + 1, fetched one chunk ahead): ping-pong
        float scv[2][8], shv[2][8];                                              // the chunk's rows of the per-row affine tables, likewise
        auto read_tab = [&](const int chunk, float (&s8)[8], float (&h8)[8]) {
            if (P_GLN || P_BWD) {
                const float4 s0 = ld4(&sm.sc[chunk * DK + 8 * kh]), s1 = ld4(&sm.sc[chunk * DK + 8 * kh + 4]);
                s8[0] = s0.x; s8[1] = s0.y; s8[2] = s0.z; s8[3] = s0.w; s8[4] = s1.x; s8[5] = s1.y; s8[6] = s1.z; s8[7] = s1.w;
            }
            if (P_GLN) {
                const float4 s0 = ld4(&sm.sh[chunk * DK + 8 * kh]), s1 = ld4(&sm.sh[chunk * DK + 8 * kh + 4]);
                h8[0] = s0.x; h8[1] = s0.y; h8[2] = s0.z; h8[3] = s0.w; h8[4] = s1.x; h8[5] = s1.y; h8[6] = s1.z; h8[7] = s1.w;
            }
        };
        auto read_raw = [&](auto stc, float (&x)[WC][8], float (&a)[WC][8]) {
            constexpr int st = decltype(stc)::value;
            const float* Xb = sm.Xr[pw][st];
            const float* Cb = sm.Cr[P_BWD ? pw : 0][P_BWD ? st : 0];
#pragma unroll
            for (int cc = 0; cc < WC; ++cc)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    x[cc][e] = Xb[r_off + 2 * PIECE * cc + 32 * e];
                    if (P_BWD) a[cc][e] = Cb[r_off + 2 * PIECE * cc + 32 * e];
                }
        };

        constexpr std::integral_constant<int, 0> Z0{};
        {
            int g = 0;
            if (xi < nk) { issue_x(std::integral_constant<int, 0>{}); ++g; }
            if (NS > 1 && xi < nk) { issue_x(std::integral_constant<int, (1 % NS)>{}); ++g; }
            if (NS > 2 && xi < nk) { issue_x(std::integral_constant<int, (2 % NS)>{}); ++g; }
            if (NS > 3 && xi < nk) { issue_x(std::integral_constant<int, (3 % NS)>{}); ++g; }
            if (NS > 4 && xi < nk) { issue_x(std::integral_constant<int, (4 % NS)>{}); ++g; }
            static_assert(NS <= 5, "prologue fill");
            // chunk 0 has landed when at most the NS - 1 newer chunks are in flight (loads only so far: in order)
            asm volatile("" ::: "memory");
            if (g == NS) __builtin_amdgcn_s_waitcnt(0x0f70 | (((NS - 1) * G) & 15) | ((((NS - 1) * G) >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0f70);
            asm volatile("" ::: "memory");
        }
        read_raw(Z0, v[0], a8[0]);
        read_tab(0, scv[0], shv[0]);

        PCD_COUNT(c_loop = -clock64());
        pcd_i4 fr = {0, 0, 0, 0};                                                // sampled freed[]: reduced when looked at
        constexpr int U = (NS * NB) / pcd_gcd(NS, NB) % 2 == 0 ? (NS * NB) / pcd_gcd(NS, NB) : 2 * (NS * NB) / pcd_gcd(NS, NB);      // even: the register ping-pong

        auto run = [&](auto fastc) {
            constexpr bool FAST = decltype(fastc)::value;                        // 0 <= alpha <= 1: PReLU(x) = max(x, alpha*x)
            auto prelu = [&](const float x) { return FAST ? pcd_vmax(x, alpha_p * x) : prelu_f(x, alpha_p); };
            auto body = [&](const int j, auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int cur = u & 1, nxt = cur ^ 1;
                constexpr int stage = u % NS, nstage = (u + 1) % NS, pb = u % NB;
                // raw chunk j is in registers: its stage may be refilled
#pragma unroll
                for (int cc = 0; cc < WC; ++cc) {
                    asm volatile("" :: "v"(v[cur][cc][0]), "v"(v[cur][cc][1]), "v"(v[cur][cc][2]), "v"(v[cur][cc][3]), "v"(v[cur][cc][4]), "v"(v[cur][cc][5]), "v"(v[cur][cc][6]), "v"(v[cur][cc][7]));
                    if (P_BWD) asm volatile("" :: "v"(a8[cur][cc][0]), "v"(a8[cur][cc][1]), "v"(a8[cur][cc][2]), "v"(a8[cur][cc][3]), "v"(a8[cur][cc][4]), "v"(a8[cur][cc][5]), "v"(a8[cur][cc][6]), "v"(a8[cur][cc][7]));
                }
                PCD_FENCE();
                // chunk j + NS goes into stage j % NS; raw chunk j + 1 is mine and has landed when only the newer chunks are in flight
                // (the GLN_BWD store-back shares vmcnt: plain vmcnt(0) there, and its refill goes behind it to have a whole
                // chunk before the next one)
                if (xi < nk) {
                    if (!P_BWD) issue_x(std::integral_constant<int, stage>{});
                    asm volatile("" ::: "memory");
                    PCD_COUNT(c_vm -= clock64());
                    if (KEEP > 0) __builtin_amdgcn_s_waitcnt(0x0f70 | (KEEP & 15) | ((KEEP >> 4) << 14));      // vmcnt only
                    else __builtin_amdgcn_s_waitcnt(0x0f70);
                    PCD_COUNT(c_vm += clock64());
                    asm volatile("" ::: "memory");
                    read_raw(std::integral_constant<int, nstage>{}, v[nxt], a8[nxt]);
                    read_tab(j + 1, scv[nxt], shv[nxt]);
                    if (P_BWD) issue_x(std::integral_constant<int, stage>{});
                } else if (j + 1 < nk) {
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    asm volatile("" ::: "memory");
                    read_raw(std::integral_constant<int, nstage>{}, v[nxt], a8[nxt]);
                    read_tab(j + 1, scv[nxt], shv[nxt]);
                }
                PCD_FENCE();

                unsigned hi[WC][4], lo[WC][4];
#pragma unroll
                for (int cc = 0; cc < WC; ++cc) {
                    float w[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = v[cur][cc][e];
                    if (P_BWD) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a = a8[cur][cc][e];
                            const float du = fmaf(w[e], scv[cur][e], fmaf(prelu(a), bk1, bk0));
                            float da = a > 0.f ? du : du * alpha_p;
                            float an = fminf(a, 0.f);
                            if (edge) { da = live[cc] ? da : 0.f; an = live[cc] ? an : 0.f; }
                            if (rt == 0) {
                                dalpha_pro = fmaf(du, an, dalpha_pro);                                            // d(alpha) += du * a where a <= 0
                                float* srow = d.pro_store + ((size_t)b * d.K + j * DK + e) * d.ldt + t0 + 128 * cc;   // uniform row pointer + lane offset
                                *reinterpret_cast<float*>(reinterpret_cast<char*>(srow) + (size_t)st_lane_off) = da;
                            }
                            w[e] = da;
                        }
                    } else if (PRO != SEP_PRO_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float x = w[e];
                            if (P_PRELU) x = prelu(x);
                            if (P_GLN) x = fmaf(x, scv[cur][e], shv[cur][e]);
                            w[e] = x;
                        }
                    }
#ifdef PCD_PROBE_NOSPLIT      // ceiling probe (tools/call_r08o.sh): X as if it arrived pre-split -- WRONG results, timing only
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hi[cc][q] = __builtin_bit_cast(unsigned, w[2 * q]); lo[cc][q] = __builtin_bit_cast(unsigned, w[2 * q + 1]); }
                    continue;
#endif
                    float m = fmaxf(fmaxf(fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fabsf(w[2])), fmaxf(fmaxf(fabsf(w[3]), fabsf(w[4])), fabsf(w[5]))),
                                    fmaxf(fabsf(w[6]), fabsf(w[7])));
                    // the column's other 8 contraction rows sit in the neighbouring lane (quad_perm [1,0,3,2])
                    m = pcd_max_quad_neighbour(m);
                    const int ex = __builtin_amdgcn_frexp_expf(m);                   // m = f * 2^ex, f in [0.5, 1)
                    const int ex2 = __builtin_bit_cast(int, m) == 0 ? -3 * UNSET : ex;     // a chunk of zeros never moves the scale
                    bexp[cc] = ex2 + bexp[cc] > 14 ? 9 - ex : bexp[cc];              // first non-zero chunk, or the column outgrew its scale
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = __builtin_ldexpf(w[e], bexp[cc]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) pcd_split2_pair(w[2 * q], w[2 * q + 1], hi[cc][q], lo[cc][q]);
                }
                // slot pb is free once every consumer has read chunk j - NB
                if (j >= NB) {
                    const int need = j - NB + 1;
                    if (pcd_min(fr) < need) {
                        PCD_COUNT(++n_miss);
                        while (pcd_min4(sm.freed) < need) { PCD_COUNT(++n_spin); __builtin_amdgcn_s_sleep(1); }
                    }
                }
                PCD_FENCE();
                float* Bp = sm.Bp[pb];
#pragma unroll
                for (int cc = 0; cc < WC; ++cc) {
                    *reinterpret_cast<u32x4_t*>(Bp + w_hi + 128 * 16 * cc) = u32x4_t{hi[cc][0], hi[cc][1], hi[cc][2], hi[cc][3]};
                    *reinterpret_cast<u32x4_t*>(Bp + w_lo + 128 * 16 * cc) = u32x4_t{lo[cc][0], lo[cc][1], lo[cc][2], lo[cc][3]};
                    sm.be[pb][colb + 128 * cc] = bexp[cc];                       // both row halves of the column store the same word
                }
                PCD_FENCE();
                pcd_post(&sm.ready[pw], j + 1);                                  // behind the operand stores in this wave's LDS order
                fr = pcd_load4(sm.freed);                                        // sampled now, looked at one chunk later
                PCD_FENCE();
            };
            for (int j0 = 0; j0 < nk; j0 += U) pcd_unroll<U>([&](auto uc) { if (j0 + decltype(uc)::value < nk) body(j0 + decltype(uc)::value, uc); });
        };
        if ((P_PRELU || P_BWD) && !(alpha_p >= 0.f && alpha_p <= 1.f)) run(std::false_type{});
        else run(std::true_type{});
        __builtin_amdgcn_s_waitcnt(0x0070);                                      // the GLN_BWD stores / nothing else is in flight
        PCD_COUNT(c_loop += clock64());
        PCD_STAMP(4, 6, n_miss); PCD_STAMP(4, 7, n_spin); PCD_STAMP(4, 8, c_loop); PCD_STAMP(4, 9, c_vm); PCD_STAMP(4, 10, wall_clock64());
    } else {
        // =================================================================================== consumer waves
        const int wr = wid / WC, wcc = wid % WC;
        const int fsw = (l31 >> 2) & 3;
        const int c_hi = l31 * 16 + 4 * ((2 * lk) ^ fsw);                        // float offset inside a 32-column block image
        const int c_lo = l31 * 16 + 4 * ((2 * lk + 1) ^ fsw);
        const int b_base = 128 * wcc * 16;
        // A fragments come straight from the packed matrix (operand-block layout: 1 KiB per 32-row block, chunk and part,
        // lane-linear), one chunk ahead of their use: L2-resident weights, perfectly coalesced, no LDS involved
        const char* Apk = reinterpret_cast<const char*>(d.A_pk) + (size_t)((m0 + 64 * wr) >> 5) * nk * 2048;     // wave-uniform
        const unsigned a_lane = 16u * (unsigned)lane;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][mi][n][r] = 0.f;
        // Operand registers: the A fragments of two consecutive chunks (sa[j & 1]) and TWO half-chunk B buffers (column blocks
        // 0-1 and 2-3): 64 operand registers beside the 128 accumulators (a full second operand set spilled).
        //
        // The instruction stream is written out by hand, one statement group per MFMA, fenced with sched_barrier: an in-order
        // wave that issues [25 other instructions | 12 MFMAs] leaves the matrix pipe idle through the first block (stamps:
        // 1460 cycles per chunk for 768 cycles of MFMAs); issued in the gaps BETWEEN the MFMAs (~7 free issue slots per 32-cycle
        // MFMA) the same instructions are free.  Two phases per chunk j:
        //   phase A: MFMAs of (chunk j-1, half 1) | ready(j)?  read B(j, half 0)                      then rescale check of half 0
        //   phase B: MFMAs of (chunk j,   half 0) | load A(j+1)  read B(j, half 1)  freed = j+1  sample ready   then check of half 1
        u32x4_t sa[2][2][2], sb[2][2][2];                                        // sa[chunk parity][mi][hi, lo]; sb[half][ni & 1][hi, lo]
        int en[2][2];                                                            // [half][ni & 1] scale exponents of the chunk's columns
        int bcur[4] = {0, 0, 0, 0};                                              // scale the accumulators of column block ni are in
        pcd_i4 rdy = {0, 0, 0, 0};                                               // sampled ready[] (reduced when looked at: the read stays in flight under the MFMAs)
#define PCD_SB() __builtin_amdgcn_sched_barrier(0)

        auto load_a1 = [&](auto parc, auto mic, auto qc, const int chunk) {      // one 16-byte fragment: (row block mi, part q) of chunk `chunk`
            constexpr int par = decltype(parc)::value, mi = decltype(mic)::value, q = decltype(qc)::value;
            const char* src = Apk + ((size_t)mi * nk + chunk) * 2048 + q * 1024;            // scalar arithmetic: the load takes the saddr form
            sa[par][mi][q] = *reinterpret_cast<const u32x4_t*>(src + a_lane);
            PCD_SB();
        };
        auto read_b1 = [&](auto halfc, auto nc, auto qc, const int pb) {         // one operand group of the split chunk in slot pb
            constexpr int h = decltype(halfc)::value, n = decltype(nc)::value, q = decltype(qc)::value;
            const float* Bb = sm.Bp[pb] + b_base + h * 64 * 16 + n * 32 * 16;
            sb[h][n][q] = *reinterpret_cast<const u32x4_t*>(Bb + (q == 0 ? c_hi : c_lo));
            PCD_SB();
        };
        // ONE address register for all exponent reads (slot, half and block go into the instruction's offset field; left to itself
        // the compiler keeps sixteen hoisted addresses and spills them)
        unsigned be_addr = lds_addr(reinterpret_cast<const float*>(&sm.be[0][128 * wcc + l31]));
        asm volatile("" : "+v"(be_addr));
        auto read_en = [&](auto halfc, const int pb) {
            constexpr int h = decltype(halfc)::value;
            en[h][0] = *(const PCD_LDS int*)(be_addr + 4u * (unsigned)(pb * TN + 64 * h));
            en[h][1] = *(const PCD_LDS int*)(be_addr + 4u * (unsigned)(pb * TN + 64 * h + 32));
            PCD_SB();
        };
        auto rescale = [&](auto halfc) {                                         // the accumulators follow their column's scale
            constexpr int h = decltype(halfc)::value;
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
            PCD_SB();
        };
        auto M = [&](auto parc, auto halfc, auto ic) {                           // MFMA i of a half chunk: parts hi*lo, lo*hi, hi*hi over the 2 x 2 blocks
            constexpr int par = decltype(parc)::value, h = decltype(halfc)::value, i = decltype(ic)::value;
            constexpr int part = i >> 2, mi = (i >> 1) & 1, n = i & 1;
            constexpr int asel = part == 1 ? 1 : 0, bsel = part == 0 ? 1 : 0;
            acc[h][mi][n] = pcd_mfma(sa[par][mi][asel], sb[h][n][bsel], acc[h][mi][n]);
            PCD_SB();
        };
        auto wait_ready = [&](const int need) {
            if (pcd_min(rdy) < need) {
                PCD_COUNT(++n_miss);
                while (pcd_min4(sm.ready) < need) { PCD_COUNT(++n_spin); __builtin_amdgcn_s_sleep(1); }
            }
            PCD_FENCE();
        };
#define PCD_I(k) std::integral_constant<int, k>{}
        // phase A of chunk j (its operands in slot pb); MF: the previous chunk's second half is multiplied meanwhile
        auto phase_a = [&](auto mfc, auto parprev, const int j, const int pb) {
            constexpr bool MF = decltype(mfc)::value;
            if (MF) M(parprev, PCD_I(1), PCD_I(0));
            wait_ready(j + 1);
            if (MF) M(parprev, PCD_I(1), PCD_I(1));
            read_b1(PCD_I(0), PCD_I(0), PCD_I(0), pb);
            if (MF) M(parprev, PCD_I(1), PCD_I(2));
            read_b1(PCD_I(0), PCD_I(0), PCD_I(1), pb);
            if (MF) M(parprev, PCD_I(1), PCD_I(3));
            read_b1(PCD_I(0), PCD_I(1), PCD_I(0), pb);
            if (MF) M(parprev, PCD_I(1), PCD_I(4));
            read_b1(PCD_I(0), PCD_I(1), PCD_I(1), pb);
            if (MF) M(parprev, PCD_I(1), PCD_I(5));
            read_en(PCD_I(0), pb);
            PCD_FENCE();
            rdy = pcd_load4(sm.ready);                                           // looked at in the next phase A: 18 MFMAs of cover
            PCD_FENCE();
            if (MF) {
                M(parprev, PCD_I(1), PCD_I(6)); M(parprev, PCD_I(1), PCD_I(7)); M(parprev, PCD_I(1), PCD_I(8));
                M(parprev, PCD_I(1), PCD_I(9)); M(parprev, PCD_I(1), PCD_I(10)); M(parprev, PCD_I(1), PCD_I(11));
            }
            rescale(PCD_I(0));
        };
        // phase B of chunk j: its first half is multiplied while A(j+1) and its second half's operands are fetched
        auto phase_b = [&](auto parc, auto parnext, const int j, const int pb) {
            const int jn = min(j + 1, nk - 1);                                   // behind the last chunk: a harmless reload of it
            M(parc, PCD_I(0), PCD_I(0));
            load_a1(parnext, PCD_I(0), PCD_I(0), jn);
            M(parc, PCD_I(0), PCD_I(1));
            load_a1(parnext, PCD_I(0), PCD_I(1), jn);
            M(parc, PCD_I(0), PCD_I(2));
            load_a1(parnext, PCD_I(1), PCD_I(0), jn);
            M(parc, PCD_I(0), PCD_I(3));
            load_a1(parnext, PCD_I(1), PCD_I(1), jn);
            M(parc, PCD_I(0), PCD_I(4));
            read_b1(PCD_I(1), PCD_I(0), PCD_I(0), pb);
            M(parc, PCD_I(0), PCD_I(5));
            read_b1(PCD_I(1), PCD_I(0), PCD_I(1), pb);
            M(parc, PCD_I(0), PCD_I(6));
            read_b1(PCD_I(1), PCD_I(1), PCD_I(0), pb);
            M(parc, PCD_I(0), PCD_I(7));
            read_b1(PCD_I(1), PCD_I(1), PCD_I(1), pb);
            M(parc, PCD_I(0), PCD_I(8));
            read_en(PCD_I(1), pb);
            M(parc, PCD_I(0), PCD_I(9));
            PCD_FENCE();
            pcd_post(&sm.freed[wid], j + 1);                                     // behind the operand reads of chunk j in this wave's LDS order
            PCD_FENCE();
            M(parc, PCD_I(0), PCD_I(10));
            M(parc, PCD_I(0), PCD_I(11));
            rescale(PCD_I(1));
        };
        // chunks per unrolled round (slot and parity become constants); the host dispatch guarantees nk % UC == 0.  The first
        // phase A multiplies zero operands (12 wasted MFMAs per tile) instead of being peeled: one loop, no second code path.
        constexpr int UC = NB % 2 == 0 ? NB : 2 * NB;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 2; ++q) { sa[1][mi][q] = u32x4_t{0u, 0u, 0u, 0u}; sb[1][mi][q] = u32x4_t{0u, 0u, 0u, 0u}; }
        load_a1(PCD_I(0), PCD_I(0), PCD_I(0), 0); load_a1(PCD_I(0), PCD_I(0), PCD_I(1), 0);
        load_a1(PCD_I(0), PCD_I(1), PCD_I(0), 0); load_a1(PCD_I(0), PCD_I(1), PCD_I(1), 0);
        PCD_COUNT(c_loop = -clock64());
        for (int j0 = 0; j0 < nk; j0 += UC)
            pcd_unroll<UC>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                phase_a(std::true_type{}, PCD_I((u & 1) ^ 1), j0 + u, u % NB);
                phase_b(PCD_I(u & 1), PCD_I((u & 1) ^ 1), j0 + u, u % NB);
            });
        pcd_unroll<12>([&](auto ic) { M(PCD_I((UC - 1) & 1), PCD_I(1), ic); });      // second half of the last chunk
        PCD_COUNT(c_loop += clock64());
        PCD_STAMP(0, 1, wall_clock64()); PCD_STAMP(0, 3, n_miss); PCD_STAMP(0, 4, n_spin); PCD_STAMP(0, 5, c_loop);
        // undo the column scales (the row scales of A leave in the epilogue)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni >> 1][mi][ni & 1][r] = __builtin_ldexpf(acc[ni >> 1][mi][ni & 1][r], -bcur[ni]);
    }
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
        gemm_epilogue<EF, 2, true, 8, true>(d, acc[h], eb, em0, et0, cw / WC, 2 * (cw % WC) + h, elane >> 5, elane & 31, etid, &sm.Xr[0][0][0], sm.red, TN, cons);
#ifdef PCD_PROF
    __builtin_amdgcn_s_waitcnt(0x0070);
    if (ewid == 0 && elane == 0 && blockIdx.x < 4096) g_pcd_prof[blockIdx.x][2] = wall_clock64();
#endif
    if (P_BWD && rt == 0) {
        const double sdal = block_sum_n<double, 8>((double)dalpha_pro, sm.red);
        if (tid == 0) atomicAdd(d.pro_dalpha, sdal);
    }
}

template <int WR, int WC, int PRO, bool SPLIT, int EF>
void launch_pcd(const sep_gemm_desc& d, hipStream_t stream) {
    constexpr bool BWD = PRO == SEP_PRO_GLN_BWD;
    // ring depths: raw ring NS (DMA prefetch distance NS - 1 chunks), operand ring NB (how far the producers may run ahead); the
    // producer loop is unrolled over lcm(NS, NB) chunks
    constexpr int NS = pcd_fits<WR, WC, 4, 4, BWD>() ? 4 : 3;
    constexpr int NB = pcd_fits<WR, WC, NS, 4, BWD>() ? 4 : 2;                  // even: the consumer's round is NB chunks
    const int NR = d.M / (64 * WR);
    const int NC = d.B * (d.ldt / (128 * WC));
    const int grid = 8 * NR * ceil_div(NC, 8);
    hipLaunchKernelGGL((pw_gemm_pc_kernel<WR, WC, PRO, SPLIT, EF, NS, NB>), dim3(grid), dim3(512), 0, stream, d);
}

}  // namespace

// Called by sep_pw_gemm_packed (gemm_coop.hip).  Returns 1 when the call was launched here.
int sep_pw_gemm_pc(const sep_gemm_desc* d, hipStream_t stream) {
    static const int force_22 = getenv("SEPK_PC_22") ? atoi(getenv("SEPK_PC_22")) : 0;
    if (d->M % 128 != 0 || d->K % (4 * DK) != 0 || d->k_split % DK != 0 || (d->m_split % 128) != 0 || d->ldt % 256 != 0) return 0;      // K: whole unrolled rounds (UC in {2, 4})
    if (d->pro_mode >= SEP_PRO_GLN && d->K > PCDMAXK) return 0;
    if ((size_t)d->M * d->K * 4 >= (1ull << 32) || (size_t)8 * d->ldt * 4 >= (1ull << 31)) return 0;     // 32-bit DMA offsets
    const int ef = d->epi_flags, pm = d->pro_mode;
    const bool sp = d->k_split != 0;
    const bool tall = d->M % 256 == 0 && !force_22;      // 256 x 128 tile (consumers 4 x 1), else 128 x 256 (2 x 2)
#define SEP_LP(P, S, E)                                          \
    do {                                                         \
        if (tall) launch_pcd<4, 1, P, S, E>(*d, stream);         \
        else launch_pcd<2, 2, P, S, E>(*d, stream);              \
        return 1;                                                \
    } while (0)
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_STATS_PRELU) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_STATS_PRELU);                      // TCN conv1
    if (!sp && pm == SEP_PRO_GLN_PRELU && ef == SEP_EPI_RESIDUAL) SEP_LP(SEP_PRO_GLN_PRELU, false, SEP_EPI_RESIDUAL);                 // heads
    if (!sp && pm == SEP_PRO_GLN_PRELU && ef == 0) SEP_LP(SEP_PRO_GLN_PRELU, false, 0);                                               // last layer: skip head only
    if (!sp && pm == SEP_PRO_PRELU && ef == SEP_EPI_SIGMOID) SEP_LP(SEP_PRO_PRELU, false, SEP_EPI_SIGMOID);                           // mask (sigmoid)
    if (!sp && pm == SEP_PRO_PRELU && ef == 0) SEP_LP(SEP_PRO_PRELU, false, 0);                                                       // mask (softmax follows)
    if (!sp && pm == SEP_PRO_GLN && ef == 0) SEP_LP(SEP_PRO_GLN, false, 0);                                                           // bottleneck
    if (!sp && pm == SEP_PRO_NONE && ef == 0) SEP_LP(SEP_PRO_NONE, false, 0);                                                         // plain 1x1 conv / input gradient
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_PRELU_BWD) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_PRELU_BWD);                         // mask^T
    if (sp && pm == SEP_PRO_NONE && ef == 0) SEP_LP(SEP_PRO_NONE, true, 0);                                                           // heads^T (its gLN sums come from the weight gradient)
    if (sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LP(SEP_PRO_NONE, true, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);    // heads^T
    if (!sp && pm == SEP_PRO_NONE && ef == (SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU)) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_ROWSUMS | SEP_EPI_ROWSUMS_PRELU);  // last layer's skip^T
    if (!sp && pm == SEP_PRO_NONE && ef == SEP_EPI_ROWSUMS) SEP_LP(SEP_PRO_NONE, false, SEP_EPI_ROWSUMS);                             // bottleneck^T
    if (!sp && pm == SEP_PRO_GLN_BWD && ef == SEP_EPI_RESIDUAL) SEP_LP(SEP_PRO_GLN_BWD, false, SEP_EPI_RESIDUAL);                     // conv1^T
    if (!sp && pm == SEP_PRO_GLN_BWD && ef == 0) SEP_LP(SEP_PRO_GLN_BWD, false, 0);
#undef SEP_LP
    return 0;
}
