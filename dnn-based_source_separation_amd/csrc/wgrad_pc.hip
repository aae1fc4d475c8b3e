// Weight gradient of the 1x1 convolutions as a PRODUCER / CONSUMER workgroup (see gemm_pc.hip for the why):
//
//   partial[s][m][n] = sum over the (sample, frame) columns of slab s of  G[b][m][t] * pro(X[b][n][t]),   partial_bias[s][m] = sum G
//
// Replaces autograd's conv weight / bias gradient of nn.Conv1d(kernel_size=1) at reference src/models/tdcn.py:86,173,175 and
// src/models/conv_tasnet.py:335,341, like pw_wgrad_split_kernel (gemm.hip), whose arithmetic it keeps: fp32 products from the
// EXACT three-way bf16 split of both operands (x = hi + mid + lo, 8 + 8 + 8 significand bits; six of the nine part products on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation) -- both operands are activations here, so there is nothing to pre-split.
//
// In pw_wgrad_split_kernel every wave splits its own fragments: 224 VALU instructions next to 24 MFMAs per 16-frame chunk, every
// G row split by 2 x N/128 waves and every X row by 2 x M/128.  Here four producer waves issue all the DMA and put each value of
// the TN x 16 X chunk through the prologue and the split ONCE, writing MFMA-ready operand planes to LDS; four consumer waves
// (64 x 128 accumulator tiles: 128 registers) read those planes with ds_read_b128, split their OWN 64 rows of G in registers
// (each G row belongs to one consumer row block, so nothing is split twice; 88 VALU in the shadow of 48 MFMAs per chunk) and
// issue the MFMAs, with the operand reads of the next quarter chunk in flight under the 12 MFMAs of the current one.  First
// version: the producers split G too -- s_memtime stamps (tools/wpc_prof.py) showed them at 2270 cycles per chunk for
// read + split + write against 1760 for the consumers' 48 MFMAs, the consumers idling half of every step at the barrier.
// One workgroup per CU, one barrier per chunk.
//
// Workgroup tile TM x TN = 256 x 128 (consumers stacked 4 x 1) or 128 x 256 (2 x 2); slabs and tiles are laid out on the grid
// exactly as in pw_wgrad_split_kernel (same `partial` / `partial_bias` contract, same sep_reduce_slabs afterwards).
#include "gemm_common.hpp"
#include <stdlib.h>
#include <type_traits>

#ifdef WPC_PROF
__device__ long long g_wpc_step[2][64][8];
#define WSTAMP(role, s) do { if (bid == 100 && (wid & 3) == 1 && lane == 0 && j < 64) g_wpc_step[role][j][s] = clock64(); } while (0)
extern "C" int sep_debug_wpc_step(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wpc_step), sizeof(long long) * 2 * 64 * 8) == hipSuccess ? 0 : -1; }
#else
#define WSTAMP(role, s) do { } while (0)
#endif

namespace {

constexpr int WPMAXB = 256;      // samples whose gLN constants fit the LDS table

template <int WR, int WC, int NS>
struct __attribute__((aligned(16))) WpcSmem {
    static constexpr int TM = 64 * WR, TN = 128 * WC;
    float Gr[NS + 1][TM * DK];      // raw G chunk [row][16 frames], 16-byte granules XOR-swizzled by ((row >> 2) & 3); one stage more
                                    // than X: the consumers read chunk j+1 while chunk j+NS is already being fetched
    float Xr[NS][TN * DK];          // raw X chunk, same layout
    float Xp[2][6][TN * 4];         // split X chunk: plane (frame half lk, part) -> [row][16 B = 8 bf16]
    float mu[WPMAXB];
    float rstd[WPMAXB];
};

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 wbf16x8_t;

__device__ __forceinline__ f32x16 wp_mfma(const u32x4_t a, const u32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8_t, a), __builtin_bit_cast(wbf16x8_t, b), c, 0, 0, 0);
}
// two fp32 values -> their (hi, mid, lo) truncated-bf16 parts packed {x0 low half, x1 high half}; hi + mid + lo == x exactly
__device__ __forceinline__ void wp_split3_pair(const float x0, const float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(v0 & 0xffff0000u), q1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}
template <int N>
__device__ __forceinline__ void wp_wait_barrier() {      // vmcnt(N) lgkmcnt(0), then the workgroup barrier
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wp_lgkm0_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int WR, int WC, int XMODE, int NS>
__global__ __launch_bounds__(512, 2) void pw_wgrad_pc_kernel(const sep_wgrad_desc d) {
    constexpr bool X_GLN = XMODE == SEP_PRO_GLN || XMODE == SEP_PRO_GLN_PRELU;
    constexpr bool X_PRELU = XMODE == SEP_PRO_PRELU || XMODE == SEP_PRO_GLN_PRELU;
    using Smem = WpcSmem<WR, WC, NS>;
    constexpr int TM = Smem::TM, TN = Smem::TN;
    constexpr int PG = TM / 64, PX = TN / 64;             // DMA pieces (16 rows x 64 B) per producer wave and chunk
    constexpr int G = PG + PX;
    constexpr int KEEP = (NS - 2) * G;
    constexpr bool PAIRS = NS == 4;                       // raw ring of four chunks, refilled two at a time (see the producer loop)
    static_assert(sizeof(Smem) <= 160 * 1024, "LDS");
    __shared__ Smem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const int lk = lane >> 5, l31 = lane & 31;

    const int ntm = d.M / TM, ntn = d.N / TN;
    const int ntiles = ntm * ntn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, jj = bid >> 3;
    const int tile = jj % ntiles;
    const int s = (jj / ntiles) * 8 + xcd;
    if (s >= d.nsplit) return;
    const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * TN;

    const int cps_t = d.ldt / DK;                  // chunks per sample
    const long chunks_total = (long)d.B * cps_t;
    const long cper = (chunks_total + d.nsplit - 1) / d.nsplit;
    const long c_begin = (long)s * cper;
    long c_end = c_begin + cper;
    if (c_end > chunks_total) c_end = chunks_total;
    const int nk = (int)(c_end > c_begin ? c_end - c_begin : 0);

    if (X_GLN) {
        for (int bx = tid; bx < d.B; bx += 512) {
            float mu, rstd;
            gln_mu_rstd(d.x_stats + (size_t)bx * SEP_STATS_SLOTS * 2, d.count, d.eps, mu, rstd);
            sm.mu[bx] = mu; sm.rstd[bx] = rstd;
        }
    }
    const float alpha_x = X_PRELU ? d.x_alpha[0] : 0.f;
    const bool do_bias = d.partial_bias != nullptr && (tile % ntn) == 0;

    f32x16 acc[2][2][2];                                 // [column half][mi][n]: the consumer's 64 x 128 tile (producers: unused)

    if (producer) {
        // =================================================================================== producer waves
        const int pw = wid - 4;
        const int ptid = tid - 256;
        // G rows of this tile live in G (rows < g_split) or G2; a 16-row DMA piece never straddles (g_split % 128 == 0)
        const int Mg1 = d.g_split ? d.g_split : d.M;
        // per-lane DMA source offsets: row r16 of the piece, granule (lane & 3) swizzled by the row
        const int r16 = lane >> 2, cch = (lane & 3) ^ ((r16 >> 2) & 3);
        unsigned voffG[PG], voffX[PX];
        const float* srcG[PG];
        int MgOf[PG];
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            const int row = m0 + 16 * (pw + 4 * q) + r16;                        // absolute row of dW = row of [G; G2]
            const bool second = d.g_split && (m0 + 16 * (pw + 4 * q)) >= d.g_split;   // wave-uniform
            srcG[q] = second ? d.G2 : d.G;
            MgOf[q] = second ? d.M - d.g_split : Mg1;
            voffG[q] = 4u * (unsigned)((row - (second ? d.g_split : 0)) * d.ldt + 4 * cch);
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) voffX[q] = 4u * (unsigned)((n0 + 16 * (pw + 4 * q) + r16) * d.ldt + 4 * cch);
        // Source pointers of the NEXT chunk to fetch, one per DMA piece, advanced by 16 frames per chunk (or to the next sample's rows): formed
        // afresh from (sample, chunk) for every piece, the 64-bit scalar multiplies made the twelve DMA instructions of a chunk pair cost
        // 1300 - 2300 cycles of one producer wave, with the consumers waiting at the barrier behind it (s_memtime stamps, tools/wpc16_prof.py)
        int it = (int)(c_begin % cps_t);
        int ci = 0, cst = 0, gst = 0;
        const float* pG[PG];
        long wrapG[PG];
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            pG[q] = srcG[q] + (size_t)(c_begin / cps_t) * MgOf[q] * d.ldt + it * DK;
            wrapG[q] = (long)MgOf[q] * d.ldt - (long)(cps_t - 1) * DK;
        }
        const float* pX = d.X + (size_t)(c_begin / cps_t) * d.N * d.ldt + it * DK;
        const long wrapX = (long)d.N * d.ldt - (long)(cps_t - 1) * DK;
        auto issue = [&]() {
#pragma unroll
            for (int q = 0; q < PG; ++q)
                glds16_asm(pG[q], voffG[q], lds_addr(&sm.Gr[gst][16 * (pw + 4 * q) * DK]));
#pragma unroll
            for (int q = 0; q < PX; ++q)
                glds16_asm(pX, voffX[q], lds_addr(&sm.Xr[cst][16 * (pw + 4 * q) * DK]));
            const bool wrap = ++it >= cps_t;                                     // the next chunk is the first of the next sample
            if (wrap) it = 0;
#pragma unroll
            for (int q = 0; q < PG; ++q) pG[q] += wrap ? wrapG[q] : (long)DK;
            pX += wrap ? wrapX : (long)DK;
            ++ci;
            cst = cst + 1 == NS ? 0 : cst + 1;
            gst = gst + 1 == NS + 1 ? 0 : gst + 1;
        };

        // split assignment: an operand with 256 rows gives every producer thread one whole row (16 frames = both frame halves),
        // one with 128 rows gives it half a row (8 frames = one frame half)
        constexpr bool X_FULL = TN == 256;
        const int x_row = X_FULL ? ptid : ptid >> 1, x_half = X_FULL ? 0 : ptid & 1;
        float xg = 0.f, xb = 0.f;
        if (X_GLN) { xg = d.x_gamma[n0 + x_row]; xb = d.x_beta[n0 + x_row]; }
        asm volatile("" :: "v"(xg), "v"(xb), "v"(alpha_x));                     // consumed before the first asm DMA
        int cb = (int)(c_begin / cps_t), ct = (int)(c_begin % cps_t);            // (sample, frame chunk) of the chunk being split

        // read 8 frames (two swizzled granules) of a row of a raw chunk
        auto read8 = [&](const float* raw, const int row, const int half, float (&v)[8]) {
            const int f = (row >> 2) & 3;
            const float4 a = ld4(raw + row * DK + 4 * ((2 * half) ^ f));
            const float4 b = ld4(raw + row * DK + 4 * ((2 * half + 1) ^ f));
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        };
        // split 8 values into the three operand planes of frame half `half`
        auto put8 = [&](float* planes, const int rows, const int row, const int half, const float (&v)[8]) {
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wp_split3_pair(v[2 * q], v[2 * q + 1], hi[q], mid[q], lo[q]);
            float* base = planes + (size_t)(3 * half) * rows * 4 + row * 4;
            *reinterpret_cast<u32x4_t*>(base) = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4_t*>(base + rows * 4) = u32x4_t{mid[0], mid[1], mid[2], mid[3]};
            *reinterpret_cast<u32x4_t*>(base + 2 * rows * 4) = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
        };

        __syncthreads();                                                         // mu / rstd table visible; no DMA in flight yet
#pragma unroll
        for (int g = 0; g < NS; ++g)
            if (ci < nk) issue();
        if (nk >= NS) wp_wait_barrier<(NS - 1) * G>();                           // B_-1: raw chunk 0 landed (the NS-1 newer ones may still fly)
        else wp_wait_barrier<0>();

        int stage = 0;
        for (int j = 0; j < nk; ++j) {
            WSTAMP(1, 0);
            const float* Xb = sm.Xr[stage];
            float* Xp = &sm.Xp[j & 1][0][0];
            float sc = 1.f, sh = 0.f;
            if (X_GLN) {
                const float rstd = sm.rstd[cb], mu = sm.mu[cb];
                sc = xg * rstd;
                sh = xb - mu * sc;
            }
            // ---- X with its prologue
#pragma unroll
            for (int h = 0; h < (X_FULL ? 2 : 1); ++h) {
                const int half = X_FULL ? h : x_half;
                float v[8];
                read8(Xb, x_row, half, v);
                if (XMODE != SEP_PRO_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = v[e];
                        if (X_PRELU) x = prelu_f(x, alpha_x);
                        v[e] = X_GLN ? x * sc + sh : x;
                    }
                }
                put8(Xp, TN, x_row, half, v);
            }
            if (++ct >= cps_t) { ct = 0; ++cb; }
            WSTAMP(1, 1);
#ifdef WPC_PROF
            __builtin_amdgcn_s_waitcnt(0x0070);
            WSTAMP(1, 2);
#endif
            // raw chunk j+1 has landed -- mine: all but the newer chunks; everyone's: the barrier -- and the operands of chunk j are written
            if (PAIRS) {
                // Chunks are fetched in PAIRS (behind the odd barriers: chunks j+3, j+4 into the stages of j-1, j): the two 64-byte pieces a
                // chunk pair takes of every 128-byte line are then requested back to back and the L2 fetches the line ONCE.  Issued one chunk
                // (~1 us) apart, the second piece found its line evicted again -- rows of the (B, C, ldt = 4096) activations are 16 KiB apart,
                // a few L2 sets for a tile's 384 rows: measured 1.9x the algorithmic bytes (331 MB for 173, rocprofv3 TCC_EA0_RDREQ; with a row
                // stride of 4224 floats the same kernel read 178; profiles/r03f_wgrad_fetch.txt).
                if (j & 1) { if (j + 3 <= nk) wp_wait_barrier<G>(); else wp_wait_barrier<0>(); }          // newer: chunk j+2
                else { if (j + 4 <= nk) wp_wait_barrier<2 * G>(); else wp_wait_barrier<0>(); }            // newer: chunks j+2, j+3
                WSTAMP(1, 3);
                if (j & 1) {
                    if (ci < nk) issue();
                    if (ci < nk) issue();
                }
            } else {
                if (KEEP > 0 && j + NS <= nk) wp_wait_barrier<KEEP>();
                else wp_wait_barrier<0>();                                       // B_j
                WSTAMP(1, 3);
                if (ci < nk) issue();                                            // chunk j+NS into the stage B_j freed
            }
            WSTAMP(1, 4);
            stage = stage + 1 == NS ? 0 : stage + 1;
        }
    } else {
        // =================================================================================== consumer waves
        const int wr = wid / WC, wcc = wid % WC;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][mi][n][r] = 0.f;
        // operand registers: the A (= G) fragments of the two 32-row blocks, split in registers from the raw DMA ring, and the
        // B (= X) fragments of the two 64-column halves from the producers' planes, each (hi, mid, lo); reads and MFMAs are staggered
        // by a quarter chunk, the raw G values of chunk j+1 are fetched at the end of step j
        u32x4_t sa[2][3], sb[2][2][3];
        float ra[2][8];                                                          // raw G: [mi][frame 8*lk + e] of row 64*wr + 32*mi + l31
        float bias_acc[2] = {0.f, 0.f};
        const int g_f = (l31 >> 2) & 3;                                          // granule swizzle of this lane's rows (any multiple of 16 added)
        const int g_off = (64 * wr + l31) * DK;                                  // + mi * 32 * DK + 4 * (granule ^ g_f)
        const int b_off = (128 * wcc + l31) * 4 + lk * 3 * TN * 4;               // + (64 * h + 32 * n) * 4 + part * TN * 4
        auto read_raw_a = [&](const int gstage) {
            const float* Gb = sm.Gr[gstage] + g_off;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float4 x = ld4(Gb + mi * 32 * DK + 4 * ((2 * lk) ^ g_f));
                const float4 y = ld4(Gb + mi * 32 * DK + 4 * ((2 * lk + 1) ^ g_f));
                ra[mi][0] = x.x; ra[mi][1] = x.y; ra[mi][2] = x.z; ra[mi][3] = x.w; ra[mi][4] = y.x; ra[mi][5] = y.y; ra[mi][6] = y.z; ra[mi][7] = y.w;
            }
        };
        auto split_a = [&](auto mic) {
            constexpr int mi = decltype(mic)::value;
            if (do_bias) bias_acc[mi] += ((ra[mi][0] + ra[mi][1]) + (ra[mi][2] + ra[mi][3])) + ((ra[mi][4] + ra[mi][5]) + (ra[mi][6] + ra[mi][7]));
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wp_split3_pair(ra[mi][2 * q], ra[mi][2 * q + 1], hi[q], mid[q], lo[q]);
            sa[mi][0] = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
            sa[mi][1] = u32x4_t{mid[0], mid[1], mid[2], mid[3]};
            sa[mi][2] = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
        };
        auto load_b = [&](auto hc, const int buf) {
            constexpr int h = decltype(hc)::value;
            const float* p = &sm.Xp[buf][0][0] + b_off + 64 * h * 4;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int part = 0; part < 3; ++part) sb[h][n][part] = *reinterpret_cast<const u32x4_t*>(p + n * 32 * 4 + part * TN * 4);
        };
        auto quarter = [&](auto mic, auto hc) {
            constexpr int mi = decltype(mic)::value, h = decltype(hc)::value;
            // x*y = hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi (the three dropped part products are <= 2^-24 |xy| each)
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[h][mi][n] = wp_mfma(sa[mi][PA[q]], sb[h][n][PB[q]], acc[h][mi][n]);
        };
        // One quarter (12 MFMAs) with the split of the G fragments of row block `ms` woven in: one pair of values (11 VALU) behind every
        // third MFMA, each step fenced with sched_barrier so that the VALU stay in the MFMAs' shadow.  Left to itself hipcc emits the
        // split as one 130-instruction block in front of the MFMAs (and sched_group_barrier did not change that): the matrix pipe
        // then idles through the block, in every chunk.
        auto quarter_and_split = [&](auto mic, auto hc, auto msc) {
            constexpr int mi = decltype(mic)::value, h = decltype(hc)::value, ms = decltype(msc)::value;
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
            unsigned hi[4], mid[4], lo[4];
            if (do_bias) bias_acc[ms] += ((ra[ms][0] + ra[ms][1]) + (ra[ms][2] + ra[ms][3])) + ((ra[ms][4] + ra[ms][5]) + (ra[ms][6] + ra[ms][7]));
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                acc[h][mi][k & 1] = wp_mfma(sa[mi][PA[k >> 1]], sb[h][k & 1][PB[k >> 1]], acc[h][mi][k & 1]);
                if (k % 3 == 0) {
                    wp_split3_pair(ra[ms][2 * (k / 3)], ra[ms][2 * (k / 3) + 1], hi[k / 3], mid[k / 3], lo[k / 3]);
                    asm volatile("" : "+v"(hi[k / 3]), "+v"(mid[k / 3]), "+v"(lo[k / 3]));      // materialise HERE (pure arithmetic is otherwise sunk past the MFMAs)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            sa[ms][0] = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
            sa[ms][1] = u32x4_t{mid[0], mid[1], mid[2], mid[3]};
            sa[ms][2] = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
        };
        constexpr std::integral_constant<int, 0> I0{};
        constexpr std::integral_constant<int, 1> I1{};
        __syncthreads();                                                         // (the producers' table barrier)
        wp_lgkm0_barrier();                                                      // B_-1: raw chunk 0 has landed
        int gstage = 0;
        if (nk > 0) read_raw_a(0);
        // chunk j after B_j: load B0 | split A0 | quarter (1,1) of chunk j-1 | split A1 | quarter (0,0) | load B1 | quarter (1,0) |
        //                    quarter (0,1) | read raw G of chunk j+1 (landed before B_j)
        if (nk > 0) {                                                            // chunk 0: nothing older to finish
            wp_lgkm0_barrier();                                                  // B_0
            load_b(I0, 0);
            split_a(I0);
            split_a(I1);
            quarter(I0, I0);
            load_b(I1, 0);
            quarter(I1, I0);
            quarter(I0, I1);
            gstage = 1;
            read_raw_a(gstage);                                                  // (chunk 1, or a stage nobody needs when nk == 1)
        }
        for (int j = 1; j < nk; ++j) {
            WSTAMP(0, 0);
            wp_lgkm0_barrier();                                                  // B_j: the X operands of chunk j are there
            WSTAMP(0, 1);
            const int buf = j & 1;
            load_b(I0, buf);
            quarter_and_split(I1, I1, I0);                                       // the last quarter of chunk j-1 || split of row block 0 of chunk j
            // sa[1] is still the old chunk's in the next call: row block 1 is split beside a quarter that uses row block 0
            quarter_and_split(I0, I0, I1);
            load_b(I1, buf);
            quarter(I1, I0);
            quarter(I0, I1);
            gstage = gstage + 1 == NS + 1 ? 0 : gstage + 1;
            read_raw_a(gstage);                                                  // raw G of chunk j+1 (landed before B_j; unused after the last chunk)
            WSTAMP(0, 2);
        }
        if (nk > 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            quarter(I1, I1);
        }
        if (do_bias && wcc == 0) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float tot = bias_acc[mi] + __shfl_xor(bias_acc[mi], 32, 64);     // the two lane halves own different frames
                if (lk == 0) {
                    d.partial_bias[(size_t)s * d.M + m0 + 64 * wr + 32 * mi + l31] = tot;
                }
            }
        }
    }
    // ---- the consumers' tiles leave for the slab.  The C layout gives a lane one column of 16 rows per accumulator, i.e. 4-byte stores
    // (128 per lane and tile: store-issue bound, like the first GEMM epilogue).  Each wave transposes 32 rows at a time through a private
    // piece of the (now idle) staging memory and stores float4: 256 contiguous bytes per 16 lanes, 4 rows per instruction.
    __syncthreads();                                    // every wave is out of the rings: the staging area becomes the transpose buffer
    if (!producer) {
        int etid = tid, es = s, em0 = m0, en0 = n0;
        asm volatile("" : "+v"(etid), "+s"(es), "+s"(em0), "+s"(en0));
        const int ewid = __builtin_amdgcn_readfirstlane(etid >> 6);
        const int ewr = ewid / WC, ewc = ewid % WC, elk = (etid >> 5) & 1, el31 = etid & 31, elane = etid & 63;
        static_assert(sizeof(Smem) >= 4 * EPI_WAVE_FLOATS * sizeof(float), "transpose buffer");
        float* Tw = reinterpret_cast<float*>(&sm) + ewid * EPI_WAVE_FLOATS;
        const int rsub = elane >> 4, c4 = elane & 15;
        float* out = d.partial + (size_t)es * d.M * d.N + (size_t)(em0 + ewr * 64) * d.N + en0 + ewc * 128 + 4 * c4;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (r & 3) + 8 * (r >> 2) + 4 * elk;
                    Tw[rl * EPI_LD + el31] = acc[h][mi][0][r];
                    Tw[rl * EPI_LD + 32 + el31] = acc[h][mi][1][r];
                }
                __builtin_amdgcn_wave_barrier();       // LDS is in-order per wave; this only pins the compiler's order
#pragma unroll
                for (int p8 = 0; p8 < 8; ++p8) {
                    const int row = 4 * p8 + rsub;
                    st4(out + (size_t)(mi * 32 + row) * d.N + h * 64, ld4(Tw + row * EPI_LD + 4 * c4));
                }
                __builtin_amdgcn_wave_barrier();
            }
    }
}

template <int WR, int WC, int XMODE>
void launch_wpc(const sep_wgrad_desc& d, hipStream_t stream) {
    const int ntiles = (d.M / (64 * WR)) * (d.N / (128 * WC));
    const int grid = 8 * ntiles * ceil_div(d.nsplit, 8);
    // raw-ring depth NS (SEPK_WPC_NS = 2 | 3 | 4 for A/B runs): 4 = chunks fetched in pairs (one HBM fetch per 128-byte line), 2 / 3 = one
    // chunk per barrier, a DMA has NS - 1 chunk periods to land
    static const int ns = getenv("SEPK_WPC_NS") ? atoi(getenv("SEPK_WPC_NS")) : 4;
    if (ns == 4) hipLaunchKernelGGL((pw_wgrad_pc_kernel<WR, WC, XMODE, 4>), dim3(grid), dim3(512), 0, stream, d);
    else if (ns == 3) hipLaunchKernelGGL((pw_wgrad_pc_kernel<WR, WC, XMODE, 3>), dim3(grid), dim3(512), 0, stream, d);
    else hipLaunchKernelGGL((pw_wgrad_pc_kernel<WR, WC, XMODE, 2>), dim3(grid), dim3(512), 0, stream, d);
}

}  // namespace

// Called by sep_pw_wgrad (gemm.hip) for the split arithmetics.  Returns 1 when the call was launched here.
int sep_pw_wgrad_pc(const sep_wgrad_desc* d, hipStream_t stream) {
    static const bool off = getenv("SEPK_WGRAD_PC") != nullptr && atoi(getenv("SEPK_WGRAD_PC")) == 0;
    if (off || d->g_mul || d->x_div != 1 || d->B > WPMAXB || d->g_split % 128 != 0) return 0;
    const bool tall = d->M % 256 == 0 && d->N % 128 == 0;
    const bool wide = d->M % 128 == 0 && d->N % 256 == 0;
    if (!tall && !wide) return 0;
    if ((size_t)d->M * d->ldt * 4 >= (1ull << 32) || (size_t)d->N * d->ldt * 4 >= (1ull << 32)) return 0;      // 32-bit DMA offsets
    if ((long)d->nsplit > (long)d->B * (d->ldt / DK)) return 0;
#define SEP_LW(XM)                                           \
    do {                                                     \
        if (tall) launch_wpc<4, 1, XM>(*d, stream);          \
        else launch_wpc<2, 2, XM>(*d, stream);               \
        return 1;                                            \
    } while (0)
    switch (d->x_mode) {
        case SEP_PRO_NONE: SEP_LW(SEP_PRO_NONE);
        case SEP_PRO_PRELU: SEP_LW(SEP_PRO_PRELU);
        case SEP_PRO_GLN: SEP_LW(SEP_PRO_GLN);
        default: SEP_LW(SEP_PRO_GLN_PRELU);
    }
#undef SEP_LW
    return 0;
}
