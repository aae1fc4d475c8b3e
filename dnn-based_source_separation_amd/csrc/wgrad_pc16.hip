// Weight gradient of the 1x1 convolutions in the SCALED TWO-PART fp16 arithmetic (SEP_ARITH_F16X3), producer / consumer workgroup:
//
//   partial[s][m][n] = sum over the (sample, frame) columns of slab s of  G[b][m][t] * pro(X[b][n][t]),   partial_bias[s][m] = sum G
//
// Replaces autograd's conv weight / bias gradient of nn.Conv1d(kernel_size=1) at reference src/models/tdcn.py:86,173,175 and
// src/models/conv_tasnet.py:335,341, like pw_wgrad_pc_kernel (wgrad_pc.hip), whose workgroup structure, slab / tile mapping and barrier
// protocol it keeps.  The operands arrive in WHOLE 128-byte lines: a DMA instruction fetches 8 rows x 32 frames -- the two 16-frame chunks
// of a PAIR -- where wgrad_pc.hip asks for 16 rows x 64 bytes per chunk; with the rows 16 KiB apart the two half-line requests of that
// form each cost an HBM fetch of the line (rocprofv3 FETCH_SIZE: 336 MB per conv1 launch for 168 MB of operands, 186 MB in this form;
// profiles/r03q_wgrad_line.txt).  What else changes is the arithmetic of the products: wgrad_pc.hip splits both operands
// EXACTLY into three bf16 parts and issues six of the nine part products (6 x 32 matrix-pipe cycles per 32x32x16 block); here
//     x 2^s = hi + lo,  hi = fp16(x 2^s) toward zero, lo = fp16(x 2^s - hi)         (11 + 11 significand bits)
//     x y = hi_x lo_y + lo_x hi_y + hi_x hi_y                                          (3 x 32 cycles)
// as in the forward / input-gradient kernels (gemm_pc.hip).  Both operands are activations, so BOTH carry run-time scales, exact powers
// of two that only ever fall:
//   * one exponent per ROW of X, kept by the producer thread(s) that own the row, handed over with every chunk (xe[]); an X row is an
//     accumulator COLUMN, owned by a lane: the lane rescales its accumulators when the exponent drops (as gemm_pc.hip does);
//   * one exponent per ROW of G, kept by the consumer lanes that load the row; a G row is an accumulator ROW, i.e. a REGISTER of every
//     lane of the wave: when it drops, the registers of that row are rescaled in all lanes (the deltas travel by ds_bpermute).
// A new maximum re-centres the row at 2^9 with head room to 2^14 (fp16 overflows at 2^16), so after the first chunks of a slab the
// rescale branches are not taken.  Values ~2^-25 below their row's running maximum lose low bits: the error is relative to |G||X| per
// output, as for fp32 accumulation, not elementwise (same model as the forward kernels; tests/test_gpu_kernels.py).
#include "gemm_common.hpp"
#include <stdlib.h>
#include <type_traits>

#ifdef WPC16_PROF
__device__ long long g_wpc_step[2][64][8];
#define W16STAMP(role, s) do { if (bid == 100 && (wid & 3) == 1 && lane == 0 && j < 64) g_wpc_step[role][j][s] = clock64(); } while (0)
extern "C" int sep_debug_wpc_step(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wpc_step), sizeof(long long) * 2 * 64 * 8) == hipSuccess ? 0 : -1; }
#else
#define W16STAMP(role, s) do { } while (0)
#endif

namespace {

constexpr int W16MAXB = 256;      // samples whose gLN constants fit the LDS table
constexpr int W16UNSET = 10000;   // exponent of a row that has only seen zeros: any first maximum "outgrows" it, ldexp(0, anything) = 0

constexpr int W16NGP = 3, W16NXP = 2;      // raw ring depths in chunk PAIRS: G is fetched two pairs ahead of its use, X one
constexpr int W16NXB = 3;                  // operand buffers: the producers run ONE CHUNK AHEAD of the chunk being multiplied (see the kernel)
template <int WR, int WC>
struct __attribute__((aligned(16))) W16Smem {
    static constexpr int TM = 64 * WR, TN = 128 * WC;
    float Gr[W16NGP][TM * 2 * DK];  // raw G chunk pair [row][32 frames = 8 granules of 16 B]; granule q of a row sits in slot q ^ w16_f8(row)
    float Xr[W16NXP][TN * 2 * DK];  // raw X chunk pair, same layout
    float Xp[W16NXB][4][TN * 4];    // split X chunk: plane (frame half lk, part hi / lo) -> [row][16 B = 8 fp16]
    int xe[W16NXB][TN];             // the rows' scale exponents, per operand buffer
    float mu[W16MAXB];
    float rstd[W16MAXB];
};

// Slot permutation of a raw row's eight granules.  ds_read_b128 is served 16 lanes at a time from a 256-byte window of banks and the rows
// are 128 bytes apart, so rows (2i, 2i+1) need f8 to be a bijection of i = 0..7 for readers of 16 consecutive rows (the consumers' G rows,
// the producers' X rows at 256 columns), and readers of 8 consecutive rows x both chunk halves (X at 128 columns: the half toggles bit 1
// of the granule index) need bits 0 and 2 of f8 to tell i = 0..3 apart: bit 0 <- row bit 1, bit 2 <- row bit 2, bit 1 <- row bit 3.
__device__ __forceinline__ int w16_f8(const int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1); }
__device__ __forceinline__ f32x16 w16_mfma(const u32x4_t a, const u32x4_t b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// W16_CEILING_PROBE (tools/call_r08n.sh only): the main loop WITHOUT its workgroup barriers -- wrong results, no hang (nothing spins): the time such
// a build takes bounds what a flag protocol instead of the barrier per chunk could gain
#ifdef W16_CEILING_PROBE
#define W16_BARRIER() do { } while (0)
#else
#define W16_BARRIER() __builtin_amdgcn_s_barrier()
#endif
template <int N>
__device__ __forceinline__ void w16_wait_barrier() {      // vmcnt(N) lgkmcnt(0), then the workgroup barrier
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 15) | ((N >> 4) << 14));
    W16_BARRIER();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void w16_lgkm0_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    W16_BARRIER();
    asm volatile("" ::: "memory");
}
// Single-instruction forms (see gemm_pc.hip): the maximum with the lane 32 away as one v_permlane32_swap (no LDS round trip in the middle
// of the MFMA stream), with the neighbouring lane as one v_max_f32_dpp, and lo = x - float(hi half) as one mixed-precision FMA.
__device__ __forceinline__ float w16_max_halves(const float m) {
    // v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second: the operands must be two REGISTERS (the
    // builtin, handed the same value twice, was given one register by hipcc and returned garbage: tools/w16_probe.hip), hence the copy in asm
    unsigned a = __builtin_bit_cast(unsigned, m), b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float w16_max_neighbour(const float m) {      // max(m, m of lane ^ 1)
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(m));
    return r;
}
__device__ __forceinline__ void w16_split2_pair(const float x0, const float x1, unsigned& hi, unsigned& lo) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(unsigned, h);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(x0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(x1), "v"(hi));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}
__device__ __forceinline__ float w16_amax8(const float (&v)[8]) {
    return fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
}
// running scale exponent of a row given the maximum of its next 16 frames: unchanged while the scaled maximum stays below 2^14 (and for a
// chunk of zeros), else re-centred so that this maximum lands in [2^8, 2^9)
__device__ __forceinline__ int w16_next_exp(const float m, const int cur) {
    const int ex = __builtin_amdgcn_frexp_expf(m);                       // m = f 2^ex, f in [0.5, 1)
    const int ex2 = __builtin_bit_cast(int, m) == 0 ? -3 * W16UNSET : ex;
    return ex2 + cur > 14 ? 9 - ex : cur;
}

// The DMA pieces of a chunk pair (8 rows x 128 B each, 4096 B apart in LDS) in ONE statement.  Issued one by one (glds16_asm) every piece
// costs six scalar issue slots -- M0 saved, set, a wait state, the load, M0 restored, the address add -- and a pointer of its own to
// advance; the SIMD issues about one instruction per four cycles whatever its kind, and the producers' ~110 scalar instructions per chunk
// were a quarter of the chunk's time (SQ counters: profiles/r05x_*).  Here M0 is saved once and advanced once per TWO pieces: the
// instruction's immediate offset (-4096) reaches the piece in front -- it moves the LDS and the global address alike, so the even pieces'
// lane offsets carry +4096 (voffc) --, and the pieces share one base pointer per source tensor (pa: pieces 0-3, pb: 4-7).
// lds1: LDS byte address of piece 1.
__device__ __forceinline__ void w16_dma_pieces8(const float* pa, const float* pb, const unsigned (&voffc)[8], unsigned lds1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %11\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %9 offset:-4096\n\tglobal_load_lds_dwordx4 %2, %9\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %9 offset:-4096\n\tglobal_load_lds_dwordx4 %4, %9\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %5, %10 offset:-4096\n\tglobal_load_lds_dwordx4 %6, %10\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %7, %10 offset:-4096\n\tglobal_load_lds_dwordx4 %8, %10\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voffc[0]), "v"(voffc[1]), "v"(voffc[2]), "v"(voffc[3]), "v"(voffc[4]), "v"(voffc[5]), "v"(voffc[6]), "v"(voffc[7]), "s"(pa), "s"(pb), "s"(lds1)
                 : "memory", "scc");
}
__device__ __forceinline__ void w16_dma_pieces4(const float* pa, const unsigned (&voffc)[4], unsigned lds1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5 offset:-4096\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:-4096\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voffc[0]), "v"(voffc[1]), "v"(voffc[2]), "v"(voffc[3]), "s"(pa), "s"(lds1) : "memory", "scc");
}

// The kernel's body.  `bid` is the workgroup's index within ITS product: the single-product kernel passes blockIdx.x, the batched kernel
// (several products of identical shape in one launch, see below) the index within the product's share of the grid.
template <int WR, int WC, int XMODE, bool G2PRE = false>
__device__ __forceinline__ void w16_body(const sep_wgrad_desc& d, const int bid) {
    constexpr bool X_GLN = XMODE == SEP_PRO_GLN || XMODE == SEP_PRO_GLN_PRELU;
    constexpr bool X_PRELU = XMODE == SEP_PRO_PRELU || XMODE == SEP_PRO_GLN_PRELU;
    using Smem = W16Smem<WR, WC>;
    constexpr int TM = Smem::TM, TN = Smem::TN;
    constexpr int PG = TM / 32, PX = TN / 32;             // DMA pieces (8 rows x 128 B) per producer wave and chunk PAIR
    constexpr int RL = 2 * DK;                            // floats per raw row
    static_assert(sizeof(Smem) <= 160 * 1024, "LDS");
    static_assert(sizeof(Smem) >= 4 * EPI_WAVE_FLOATS * sizeof(float), "transpose buffer");
    __shared__ Smem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const int lk = lane >> 5, l31 = lane & 31;

    const int ntm = d.M / TM, ntn = d.N / TN;
    const int ntiles = ntm * ntn;
    const int xcd = bid & 7, jj = bid >> 3;
    const int tile = jj % ntiles;
    const int s = (jj / ntiles) * 8 + xcd;
    if (s >= d.nsplit) return;
    const int m0 = (tile / ntn) * TM, n0 = (tile % ntn) * TN;

    const int cps_t = d.ldt / DK;                  // chunks per sample
    const long chunks_total = (long)d.B * cps_t;
    long cper = (chunks_total + d.nsplit - 1) / d.nsplit;
    cper += cper & 1;                              // whole pairs (slab boundaries on 32-frame marks, as in wgrad_pc.hip; ldt % 32 == 0)
    const long c_begin = (long)s * cper;
    long c_end = c_begin + cper;
    if (c_end > chunks_total) c_end = chunks_total;
    const int nk = (int)(c_end > c_begin ? c_end - c_begin : 0);      // even
    const int np = nk >> 1;

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
        const int Mg1 = d.g_split ? d.g_split : d.M;
        const int r8 = lane >> 3, slot = lane & 7;       // a DMA instruction: 8 rows x 8 granules, lane-linear in LDS = [row][32 frames]
        static_assert(PG == 8 && PX == 4, "w16_dma_pieces8 / 4");
        unsigned voffG[PG], voffX[PX];                                           // (the even pieces' carry +4096: see w16_dma_pieces8)
        // g_split is a multiple of 128 and m0 of 256: pieces 0-3 (rows < 128 of the tile) and pieces 4-7 each come from ONE tensor
        const bool secA = d.g_split && m0 >= d.g_split, secB = d.g_split && m0 + 128 >= d.g_split;
        const int MgA = secA ? d.M - d.g_split : Mg1, MgB = secB ? d.M - d.g_split : Mg1;
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            const int rowt = 8 * (pw + 4 * q) + r8;                                     // row of the tile
            const bool second = q < 4 ? secA : secB;
            voffG[q] = 4u * (unsigned)((m0 + rowt - (second ? d.g_split : 0)) * d.ldt + 4 * (slot ^ w16_f8(rowt))) + ((q & 1) ? 0u : 4096u);
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int rowt = 8 * (pw + 4 * q) + r8;
            voffX[q] = 4u * (unsigned)((n0 + rowt) * d.ldt + 4 * (slot ^ w16_f8(rowt))) + ((q & 1) ? 0u : 4096u);
        }
        // Source pointers of the NEXT pair to fetch, one per source tensor, advanced by 32 frames per pair (or to the next sample's rows).
        // G runs one pair ahead of X (three raw G stages, two raw X stages).
        const int pps = cps_t / 2;                                               // pairs per sample
        int itG = (int)(c_begin % cps_t) / 2, itX = itG;
        int ciG = 0, ciX = 0, gst = 0, xst = 0;
        const float* pGa = (secA ? d.G2 : d.G) + (size_t)(c_begin / cps_t) * MgA * d.ldt + itG * RL;
        const float* pGb = (secB ? d.G2 : d.G) + (size_t)(c_begin / cps_t) * MgB * d.ldt + itG * RL;
        const long wrapGa = (long)MgA * d.ldt - (long)(pps - 1) * RL, wrapGb = (long)MgB * d.ldt - (long)(pps - 1) * RL;
        const float* pX = d.X + (size_t)(c_begin / cps_t) * d.N * d.ldt + itX * RL;
        const long wrapX = (long)d.N * d.ldt - (long)(pps - 1) * RL;
        auto issueG = [&]() {
            w16_dma_pieces8(pGa, pGb, voffG, lds_addr(&sm.Gr[gst][8 * (pw + 4) * RL]));
            const bool wrap = ++itG >= pps;                                      // the next pair is the first of the next sample
            if (wrap) itG = 0;
            pGa += wrap ? wrapGa : (long)RL;
            pGb += wrap ? wrapGb : (long)RL;
            ++ciG;
            gst = gst + 1 == W16NGP ? 0 : gst + 1;
        };
        auto issueX = [&]() {
            w16_dma_pieces4(pX, voffX, lds_addr(&sm.Xr[xst][8 * (pw + 4) * RL]));
            const bool wrap = ++itX >= pps;
            if (wrap) itX = 0;
            pX += wrap ? wrapX : (long)RL;
            ++ciX;
            xst = xst + 1 == W16NXP ? 0 : xst + 1;
        };

        // an operand with 256 rows gives every producer thread one whole row (16 frames), one with 128 rows half a row (8 frames): the
        // row's two halves then sit in neighbouring lanes
        constexpr bool X_FULL = TN == 256;
        const int x_row = X_FULL ? ptid : ptid >> 1, x_half = X_FULL ? 0 : ptid & 1;
        float xg = 0.f, xb = 0.f;
        if (X_GLN) { xg = d.x_gamma[n0 + x_row]; xb = d.x_beta[n0 + x_row]; }
        asm volatile("" :: "v"(xg), "v"(xb), "v"(alpha_x));
        int cb = (int)(c_begin / cps_t), ct = (int)(c_begin % cps_t);
        int xexp = W16UNSET;                                                     // this row's running scale exponent

        auto read8 = [&](const float* raw, const int par, const int row, const int half, float (&v)[8]) {      // chunk `par` of the pair
            const int f = w16_f8(row);
            const float4 a = ld4(raw + row * RL + 4 * ((4 * par + 2 * half) ^ f));
            const float4 b = ld4(raw + row * RL + 4 * ((4 * par + 2 * half + 1) ^ f));
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        };
        auto put8 = [&](float* planes, const int row, const int half, const float (&v)[8]) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w16_split2_pair(v[2 * q], v[2 * q + 1], hi[q], lo[q]);
            float* base = planes + (size_t)(2 * half) * TN * 4 + row * 4;
            *reinterpret_cast<u32x4_t*>(base) = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u32x4_t*>(base + TN * 4) = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
        };

        // operands of chunk i: this thread's row (half row) of X scaled, split and written to buffer i % 3
        auto process = [&](const int i) {
            const int j = i;                                                     // (the stamps' chunk index)
            const int P = i >> 1, par = i & 1;
            W16STAMP(1, 0);
            const float* Xb = sm.Xr[P & 1];
            const int xb3 = i % W16NXB;
            float* Xp = &sm.Xp[xb3][0][0];
            float sc = 1.f, sh = 0.f;
            if (X_GLN) {
                const float rstd = sm.rstd[cb], mu = sm.mu[cb];
                sc = xg * rstd;
                sh = xb - mu * sc;
            }
            float v[X_FULL ? 2 : 1][8];
            float m = 0.f;
#pragma unroll
            for (int h = 0; h < (X_FULL ? 2 : 1); ++h) {
                read8(Xb, par, x_row, X_FULL ? h : x_half, v[h]);
                if (XMODE != SEP_PRO_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = v[h][e];
                        if (X_PRELU) x = prelu_f(x, alpha_x);
                        v[h][e] = X_GLN ? x * sc + sh : x;
                    }
                }
                m = fmaxf(m, w16_amax8(v[h]));
            }
            if (!X_FULL) m = w16_max_neighbour(m);                     // the row's other eight frames sit in the neighbouring lane
            xexp = w16_next_exp(m, xexp);
#pragma unroll
            for (int h = 0; h < (X_FULL ? 2 : 1); ++h) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[h][e] = __builtin_ldexpf(v[h][e], xexp);
                put8(Xp, x_row, X_FULL ? h : x_half, v[h]);
            }
            if (X_FULL || x_half == 0) sm.xe[xb3][x_row] = xexp;                 // (both halves of a row hold the same exponent)
            if (++ct >= cps_t) { ct = 0; ++cb; }
            W16STAMP(1, 1);
#ifdef WPC16_PROF
            __builtin_amdgcn_s_waitcnt(0xc07f);
            W16STAMP(1, 2);
#endif
        };

        // The producers stay ONE CHUNK AHEAD: barrier B_j finds the operands of chunk j + 1 written (three operand buffers), so a consumer
        // fetches the first half of its next chunk's operands under the second half of this chunk's MFMAs and starts behind the barrier
        // without an LDS round trip in front of its first MFMA (stamps with operands of chunk j published AT B_j: 1530 ticks per consumer
        // step for 768 of MFMAs, 370 waiting at the barrier; profiles/r03s_wpc16_stamps.txt).  B_j also finds the raw pairs holding the
        // chunks up to j + 2 landed: the consumers read raw G of chunk j + 1 in step j, the producers raw X of chunk j + 2 behind B_j.
        __syncthreads();                                                         // mu / rstd table visible; no DMA in flight yet
        if (np > 0) { issueG(); issueX(); }
        if (np > 1) { issueG(); issueX(); }
        if (np > 2) issueG();
        if (np > 2) w16_wait_barrier<2 * PG + PX>();                             // B_-2: raw pair 0 landed (behind it: G1 X1 G2)
        else w16_wait_barrier<0>();
        if (nk > 0) process(0);
        w16_lgkm0_barrier();                                                     // B_-1: the operands of chunk 0

        for (int j = 0; j < nk; ++j) {
            const int P = j >> 1;
            if (j + 1 < nk) process(j + 1);
            if (!(j & 1)) {
                // chunk j + 2 opens pair P + 1: landed -- mine: all but the G pair fetched after it; everyone's: the barrier.  The X stage
                // of pair P has been read for the last time (chunk j + 1, just now) and takes the pair after next.
                if (P + 2 < np) w16_wait_barrier<PG>(); else w16_wait_barrier<0>();
                W16STAMP(1, 3);
                if (P + 2 < np) issueX();
            } else {
                w16_lgkm0_barrier();
                W16STAMP(1, 3);
                if (P + 3 < np) issueG();                                       // (raw G of pair P was read for the last time in step j - 1)
            }
            W16STAMP(1, 4);
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
        // Operand registers: A (= G) of the chunk being multiplied and of the NEXT chunk (sa[parity]: the next chunk is scaled and split in
        // the shadow of this chunk's MFMAs, one micro-step behind every second MFMA, each pinned with sched_barrier -- left to itself hipcc
        // emits the ~110 VALU of the split as one block in front of the MFMAs and the matrix pipe idles through it); B (= X) of both column
        // halves (sb[h]: all reads of a chunk are issued right behind its barrier).
        u32x4_t sa[2][2][2], sb[2][2][2];                                        // sa[parity][mi][hi, lo]; sb[h][n][hi, lo]
        float ra[2][8];                                                          // raw G: [mi][frame 8*lk + e] of row 64*wr + 32*mi + l31
        float bias_acc[2] = {0.f, 0.f};
        int gexp[2] = {W16UNSET, W16UNSET};                                      // running scale exponents of this lane's two G rows
        if (G2PRE) gexp[1] = d.g2_exps[(size_t)(c_begin / cps_t) * (d.M - d.g_split) + (m0 + 128 - d.g_split) + 32 * wr + l31];      // (the slab lies in ONE sample: checked by the launcher)
        int gdelta[2] = {0, 0};                                                  // their change with the chunk split last: applied to the accumulators
                                                                                 // before that chunk's first MFMA
        int bcur[2][2] = {{W16UNSET, W16UNSET}, {W16UNSET, W16UNSET}};           // scale the accumulators of column block (h, n) are in
        int en[2][2];
        const int g_f = w16_f8(l31);
        // Row blocks of the 256-row tile by wave: block mi = 0 of wave wr is rows 32 wr + [0, 32), block mi = 1 rows 128 + 32 wr + [0, 32) -- with a two-source G
        // (g_split = 128: [dout; dS] of the heads) every wave owns 32 rows of each source, so that a PRE-SPLIT second source (G2PRE) halves the split
        // arithmetic of EVERY consumer wave instead of freeing two of the four
        const int g_off = (32 * wr + l31) * RL;
        const int b_off = (128 * wcc + l31) * 4 + lk * 2 * TN * 4;               // + (64 * h + 32 * n) * 4 + part * TN * 4
#define W16_SB() __builtin_amdgcn_sched_barrier(0)
        // raw G rows of block mi of chunk `par` of the pair in stage gstage
        auto read_raw_mi = [&](const int gstage, auto parc, auto mic) __attribute__((always_inline)) {
            constexpr int par = decltype(parc)::value, mi = decltype(mic)::value;
            const float* Gb = sm.Gr[gstage] + g_off + mi * 128 * RL;
            // raw fp32: frames 8 lk .. 8 lk + 7 of chunk `par` are granules 4 par + 2 lk, + 1 of the row's 128-byte line; a pre-split line holds
            // {32 hi | 32 lo} fp16: the eight hi values are granule 2 par + lk, the eight lo values granule 4 + 2 par + lk
            constexpr bool pre = G2PRE && mi == 1;
            const float4 x = ld4(Gb + 4 * ((pre ? 2 * par + lk : 4 * par + 2 * lk) ^ g_f));
            const float4 y = ld4(Gb + 4 * ((pre ? 4 + 2 * par + lk : 4 * par + 2 * lk + 1) ^ g_f));
            ra[mi][0] = x.x; ra[mi][1] = x.y; ra[mi][2] = x.z; ra[mi][3] = x.w; ra[mi][4] = y.x; ra[mi][5] = y.y; ra[mi][6] = y.z; ra[mi][7] = y.w;
            W16_SB();
        };
        // micro-steps of the split of row block mi: (A) the row's new scale exponent; (B q) values 2q, 2q+1 scaled and split
        auto split_exp = [&](auto mic, const bool live) __attribute__((always_inline)) {
            constexpr int mi = decltype(mic)::value;
            if constexpr (G2PRE && mi == 1) { (void)live; gdelta[mi] = 0; W16_SB(); return; }      // one scale per row and sample, fixed by sep_split_rows
#if defined(W16_PROBE_NOSPLIT) || defined(W16_PROBE_NOSPLIT_MI1)      // ceiling probes (tools/call_r08o.sh, r08r): G (or the second row block of every wave) as if it arrived pre-split -- WRONG results, timing only
#ifdef W16_PROBE_NOSPLIT_MI1
            if (mi == 1)
#endif
            { (void)live; gdelta[mi] = 0; W16_SB(); return; }
#endif
            const float rsum = ((ra[mi][0] + ra[mi][1]) + (ra[mi][2] + ra[mi][3])) + ((ra[mi][4] + ra[mi][5]) + (ra[mi][6] + ra[mi][7]));
            bias_acc[mi] += (do_bias && live) ? rsum : 0.f;
            float m = w16_amax8(ra[mi]);
            m = w16_max_halves(m);                                               // the row's other eight frames: lane + 32
            const int nexp = live ? w16_next_exp(m, gexp[mi]) : gexp[mi];
            gdelta[mi] = nexp - gexp[mi];
            gexp[mi] = nexp;
            asm volatile("" : "+v"(gdelta[mi]), "+v"(gexp[mi]));                 // materialise HERE (pure arithmetic is otherwise sunk past the MFMAs)
            W16_SB();
        };
        auto split_pair = [&](auto parc, auto mic, auto qc) __attribute__((always_inline)) {      // straight into the operand registers of set `par`
            constexpr int par = decltype(parc)::value, mi = decltype(mic)::value, q = decltype(qc)::value;
            unsigned hi, lo;
            if constexpr (G2PRE && mi == 1) {                                    // the registers hold the operand halves already
                sa[par][mi][0][q] = __builtin_bit_cast(unsigned, ra[mi][q]);
                sa[par][mi][1][q] = __builtin_bit_cast(unsigned, ra[mi][4 + q]);
                asm volatile("" : "+v"(sa[par][mi][0]), "+v"(sa[par][mi][1]));
                W16_SB();
                return;
            }
#if defined(W16_PROBE_NOSPLIT)
            hi = __builtin_bit_cast(unsigned, ra[mi][2 * q]); lo = __builtin_bit_cast(unsigned, ra[mi][2 * q + 1]);
#elif defined(W16_PROBE_NOSPLIT_MI1)
            if (mi == 1) { hi = __builtin_bit_cast(unsigned, ra[mi][2 * q]); lo = __builtin_bit_cast(unsigned, ra[mi][2 * q + 1]); }
            else w16_split2_pair(__builtin_ldexpf(ra[mi][2 * q], gexp[mi]), __builtin_ldexpf(ra[mi][2 * q + 1], gexp[mi]), hi, lo);
#else
            w16_split2_pair(__builtin_ldexpf(ra[mi][2 * q], gexp[mi]), __builtin_ldexpf(ra[mi][2 * q + 1], gexp[mi]), hi, lo);
#endif
            sa[par][mi][0][q] = hi;
            sa[par][mi][1][q] = lo;
            asm volatile("" : "+v"(sa[par][mi][0]), "+v"(sa[par][mi][1]));
            W16_SB();
        };
        // when a G row's maximum outgrew its scale the accumulator ROWS follow: a G row is register r of EVERY lane (rare after the first chunks)
        auto follow_rows = [&]() __attribute__((always_inline)) {
            if (__builtin_amdgcn_ballot_w64((gdelta[0] | gdelta[1]) != 0) != 0) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = __shfl(gdelta[mi], (r & 3) + 8 * (r >> 2) + 4 * lk, 64);      // row of register r (C layout), kept by lane = row
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int n = 0; n < 2; ++n) acc[h][mi][n][r] = __builtin_ldexpf(acc[h][mi][n][r], dr);
                        W16_SB();                                                // one row at a time: 32 permutes hoisted to the top would spill
                    }
            }
            W16_SB();
        };
        // the operands of column half h in buffer buf
        auto load_b = [&](const int buf, auto hc) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
            const float* p = &sm.Xp[buf][0][0] + b_off;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                sb[h][n][0] = *reinterpret_cast<const u32x4_t*>(p + (64 * h + 32 * n) * 4);
                sb[h][n][1] = *reinterpret_cast<const u32x4_t*>(p + (64 * h + 32 * n) * 4 + TN * 4);
                en[h][n] = sm.xe[buf][128 * wcc + 64 * h + 32 * n + l31];
            }
            W16_SB();
        };
        // ... and the accumulator COLUMNS follow their X rows' scales (a column is a lane's own)
        auto follow_cols = [&](auto hc) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int dl = en[h][n] - bcur[h][n];
                if (__builtin_amdgcn_ballot_w64(dl != 0) != 0) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[h][mi][n][r] = __builtin_ldexpf(acc[h][mi][n][r], dl);
                }
                bcur[h][n] = en[h][n];
            }
            W16_SB();
        };
        // MFMA k of column half h: parts hi*lo, lo*hi, hi*hi over the 2 x 2 blocks (the dropped lo*lo is <= 2^-22 |xy|)
        auto M = [&](auto parc, auto hc, auto kc) __attribute__((always_inline)) {
            constexpr int par = decltype(parc)::value, h = decltype(hc)::value, k = decltype(kc)::value;
            constexpr int part = k >> 2, mi = (k >> 1) & 1, n = k & 1;
            acc[h][mi][n] = w16_mfma(sa[par][mi][part == 1 ? 1 : 0], sb[h][n][part == 0 ? 1 : 0], acc[h][mi][n]);
            W16_SB();
        };
#define W16_I(k) std::integral_constant<int, (k)>{}
        // one chunk: its 24 MFMAs with the split of the NEXT chunk's G rows woven in (more = there is a next chunk)
        // `more` = there is a next chunk.  It only feeds selects: behind the last chunk the split runs on the stage's stale contents (in-bounds,
        // never multiplied) and the rows' exponents / bias sums are left alone -- ten `last chunk?` branches per chunk in the MFMA stream, or
        // separate code paths for the tail (tried: the allocator then spills 870 registers at the joins), cost more.
        // The first column half's operands were fetched under the previous step (buffer buf), the next chunk's (buffer buf_next) are
        // fetched here as soon as this chunk's first half is multiplied.  Likewise the raw G rows: the registers hold the NEXT chunk's rows
        // when the step starts (they are split during it) and are refilled with the rows of the chunk after next (same parity, stage
        // gstage2) as soon as each row block's split is done -- no LDS round trip in front of the first split either.
        auto step = [&](auto parc, const bool more, const int buf, const int buf_next, const int gstage2) __attribute__((always_inline)) {
            constexpr int P = decltype(parc)::value;
            load_b(buf, W16_I(1));
            W16_SB();
            follow_rows();                                                       // (scales chosen while the previous chunk was multiplied)
            follow_cols(W16_I(0));
            M(W16_I(P), W16_I(0), W16_I(0)); M(W16_I(P), W16_I(0), W16_I(1));
            split_exp(W16_I(0), more);
            M(W16_I(P), W16_I(0), W16_I(2)); M(W16_I(P), W16_I(0), W16_I(3));
            split_pair(W16_I(P ^ 1), W16_I(0), W16_I(0));
            M(W16_I(P), W16_I(0), W16_I(4)); M(W16_I(P), W16_I(0), W16_I(5));
            split_pair(W16_I(P ^ 1), W16_I(0), W16_I(1));
            M(W16_I(P), W16_I(0), W16_I(6)); M(W16_I(P), W16_I(0), W16_I(7));
            split_pair(W16_I(P ^ 1), W16_I(0), W16_I(2));
            M(W16_I(P), W16_I(0), W16_I(8)); M(W16_I(P), W16_I(0), W16_I(9));
            split_pair(W16_I(P ^ 1), W16_I(0), W16_I(3));
            read_raw_mi(gstage2, W16_I(P), W16_I(0));
            M(W16_I(P), W16_I(0), W16_I(10)); M(W16_I(P), W16_I(0), W16_I(11));
            load_b(buf_next, W16_I(0));                                          // (behind the last chunk: a stale buffer, never multiplied)
            follow_cols(W16_I(1));
            M(W16_I(P), W16_I(1), W16_I(0)); M(W16_I(P), W16_I(1), W16_I(1));
            split_exp(W16_I(1), more);
            M(W16_I(P), W16_I(1), W16_I(2)); M(W16_I(P), W16_I(1), W16_I(3));
            split_pair(W16_I(P ^ 1), W16_I(1), W16_I(0));
            M(W16_I(P), W16_I(1), W16_I(4)); M(W16_I(P), W16_I(1), W16_I(5));
            split_pair(W16_I(P ^ 1), W16_I(1), W16_I(1));
            M(W16_I(P), W16_I(1), W16_I(6)); M(W16_I(P), W16_I(1), W16_I(7));
            split_pair(W16_I(P ^ 1), W16_I(1), W16_I(2));
            M(W16_I(P), W16_I(1), W16_I(8)); M(W16_I(P), W16_I(1), W16_I(9));
            split_pair(W16_I(P ^ 1), W16_I(1), W16_I(3));
            read_raw_mi(gstage2, W16_I(P), W16_I(1));
            M(W16_I(P), W16_I(1), W16_I(10)); M(W16_I(P), W16_I(1), W16_I(11));
            W16_SB();
        };
        __syncthreads();                                                         // (the producers' table barrier)
        w16_lgkm0_barrier();                                                     // B_-2: raw pair 0 has landed
        int gstage = 0;
        if (nk > 0) {                                                            // chunk 0 is scaled and split up front
            read_raw_mi(0, W16_I(0), W16_I(0)); read_raw_mi(0, W16_I(0), W16_I(1));
            split_exp(W16_I(0), true); split_exp(W16_I(1), true);
            split_pair(W16_I(0), W16_I(0), W16_I(0)); split_pair(W16_I(0), W16_I(0), W16_I(1)); split_pair(W16_I(0), W16_I(0), W16_I(2)); split_pair(W16_I(0), W16_I(0), W16_I(3));
            split_pair(W16_I(0), W16_I(1), W16_I(0)); split_pair(W16_I(0), W16_I(1), W16_I(1)); split_pair(W16_I(0), W16_I(1), W16_I(2)); split_pair(W16_I(0), W16_I(1), W16_I(3));
            read_raw_mi(0, W16_I(1), W16_I(0)); read_raw_mi(0, W16_I(1), W16_I(1));      // chunk 1 (same pair): split during step 0
        }
        w16_lgkm0_barrier();                                                     // B_-1: the X operands of chunk 0 are there
        load_b(0, W16_I(0));
        int buf = 0;
        for (int j = 0; j < nk; j += 2) {
            W16STAMP(0, 0);
            w16_lgkm0_barrier();                                                 // B_j: the X operands of chunk j+1 are there, the raw pair of chunk j+2 too
            W16STAMP(0, 1);
            const int gnext = gstage + 1 == W16NGP ? 0 : gstage + 1;            // stage of the NEXT pair: chunks j+2 and j+3
            int nb = buf + 1 == W16NXB ? 0 : buf + 1;
            step(W16_I(0), j + 1 < nk, buf, nb, gnext);
            buf = nb;
            W16STAMP(0, 2);
            if (j + 1 < nk) {
                w16_lgkm0_barrier();                                             // B_{j+1}
                W16STAMP(0, 3);
                nb = buf + 1 == W16NXB ? 0 : buf + 1;
                step(W16_I(1), j + 2 < nk, buf, nb, gnext);
                buf = nb;
                W16STAMP(0, 4);
            }
            gstage = gnext;
        }
        follow_rows();                                                           // (a change decided with the last split has nothing to follow: no-op)
        // undo the scales: accumulator (row, column) is in units of 2^(gexp[row] + bcur[column])
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = __shfl(gexp[mi], (r & 3) + 8 * (r >> 2) + 4 * lk, 64);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[h][mi][n][r] = __builtin_ldexpf(acc[h][mi][n][r], -(gr + bcur[h][n]));
                __builtin_amdgcn_sched_barrier(0);
            }
        if (do_bias && wcc == 0) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                float tot = bias_acc[mi] + __shfl_xor(bias_acc[mi], 32, 64);           // the two lane halves own different frames
                if (G2PRE && mi == 1) tot = d.g2_sums[(size_t)s * (d.M - d.g_split) + (m0 + 128 - d.g_split) + 32 * wr + l31];      // summed once by sep_split_rows
                if (lk == 0) d.partial_bias[(size_t)s * d.M + m0 + 128 * mi + 32 * wr + l31] = tot;
            }
        }
    }
    // ---- the consumers' tiles leave for the slab: transposed through LDS, float4 stores (see wgrad_pc.hip)
    __syncthreads();
    if (!producer) {
        int etid = tid, es = s, em0 = m0, en0 = n0;
        asm volatile("" : "+v"(etid), "+s"(es), "+s"(em0), "+s"(en0));
        const int ewid = __builtin_amdgcn_readfirstlane(etid >> 6);
        const int ewr = ewid / WC, ewc = ewid % WC, elk = (etid >> 5) & 1, el31 = etid & 31, elane = etid & 63;
        float* Tw = reinterpret_cast<float*>(&sm) + ewid * EPI_WAVE_FLOATS;
        const int rsub = elane >> 4, c4 = elane & 15;
        float* out = d.partial + (size_t)es * d.M * d.N + (size_t)(em0 + ewr * 32) * d.N + en0 + ewc * 128 + 4 * c4;
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
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int p8 = 0; p8 < 8; ++p8) {
                    const int row = 4 * p8 + rsub;
                    st4(out + (size_t)(mi * 128 + row) * d.N + h * 64, ld4(Tw + row * EPI_LD + 4 * c4));
                }
                __builtin_amdgcn_wave_barrier();
            }
    }
}

template <int WR, int WC, int XMODE, bool G2PRE = false>
__global__ __launch_bounds__(512, 2) void pw_wgrad_pc16_kernel(const sep_wgrad_desc d) {
    w16_body<WR, WC, XMODE, G2PRE>(d, (int)blockIdx.x);
}

// Several weight gradients of IDENTICAL shape and prologue in one launch (sep_pw_wgrad_batch): the conv1 weight gradients of consecutive TCN
// layers are leaves of the backward pass -- nothing waits for them -- so the host holds them back and issues L of them together.  The
// grid stays one workgroup per compute unit, so every product gets 1/L of the slabs: L times longer contractions per workgroup (the
// first-chunk prologue and the 128 KiB slab store of a workgroup are paid once per L times as many chunks) and 1/L of the slab
// traffic behind them (written here, read by sep_reduce_slabs).  Workgroups [k * per, (k + 1) * per) work on product k.
constexpr int W16_MAXBATCH = 8;
struct W16Batch {
    sep_wgrad_desc d;                       // the common shape / prologue; its operand pointers are replaced per product
    int per;                                // workgroups per product (a multiple of 8: the XCD decode of the body sees bid & 7 = blockIdx & 7)
    const float* G[W16_MAXBATCH];
    const float* G2[W16_MAXBATCH];
    const float* X[W16_MAXBATCH];
    float* partial[W16_MAXBATCH];
    float* partial_bias[W16_MAXBATCH];
};
template <int WR, int WC, int XMODE>
__global__ __launch_bounds__(512, 2) void pw_wgrad_pc16_batch_kernel(const W16Batch b) {
    const int k = (int)blockIdx.x / b.per;
    sep_wgrad_desc d = b.d;
    d.G = b.G[k]; d.G2 = b.G2[k]; d.X = b.X[k]; d.partial = b.partial[k]; d.partial_bias = b.partial_bias[k];
    w16_body<WR, WC, XMODE>(d, (int)blockIdx.x - k * b.per);
}

template <int WR, int WC, int XMODE, bool G2PRE = false>
void launch_w16(const sep_wgrad_desc& d, hipStream_t stream) {
    const int ntiles = (d.M / (64 * WR)) * (d.N / (128 * WC));
    const int grid = 8 * ntiles * ceil_div(d.nsplit, 8);
    hipLaunchKernelGGL((pw_wgrad_pc16_kernel<WR, WC, XMODE, G2PRE>), dim3(grid), dim3(512), 0, stream, d);
}

template <int WR, int WC, int XMODE>
void launch_w16_batch(const W16Batch& b, int n, hipStream_t stream) {
    hipLaunchKernelGGL((pw_wgrad_pc16_batch_kernel<WR, WC, XMODE>), dim3(n * b.per), dim3(512), 0, stream, b);
}

}  // namespace

// Called by sep_pw_wgrad (gemm.hip) for SEP_ARITH_F16X3.  Returns 1 when the call was launched here (same shapes as sep_pw_wgrad_pc).
int sep_pw_wgrad_pc16(const sep_wgrad_desc* d, hipStream_t stream) {
    static const bool off = getenv("SEPK_WGRAD_F16") != nullptr && atoi(getenv("SEPK_WGRAD_F16")) == 0;
    if (off || d->arith != SEP_ARITH_F16X3 || d->g_mul || d->x_div != 1 || d->B > W16MAXB || d->g_split % 128 != 0) return 0;
    // 256 x 128 workgroup tiles only: with three operand buffers the 128 x 256 form does not fit the LDS; such shapes (one launch of the
    // Conv-TasNet step: the bottleneck's 128 x 512) go to the exact bf16 kernel (wgrad_pc.hip)
    if (d->M % 256 != 0 || d->N % 128 != 0) return 0;
    if ((size_t)d->M * d->ldt * 4 >= (1ull << 32) || (size_t)d->N * d->ldt * 4 >= (1ull << 32)) return 0;      // 32-bit DMA offsets
    if ((long)d->nsplit > (long)d->B * (d->ldt / DK)) return 0;
    // second G source handed over PRE-SPLIT (sep_split_rows): the heads' [dout; dS] with dS split once per step for all layers.  One 256-row tile
    // (M = 256, g_split = 128), slabs inside one sample (the row scales are per sample), the bias partials of those rows come with it.
    if (d->G2_pre != nullptr) {
        const long cps_t = d->ldt / DK, chunks_total = (long)d->B * cps_t;
        long cper = (chunks_total + d->nsplit - 1) / d->nsplit;
        cper += cper & 1;
        if (!(d->M == 256 && d->g_split == 128 && d->g2_exps && (d->g2_sums || !d->partial_bias) && cps_t % cper == 0 &&
              (d->x_mode == SEP_PRO_PRELU || d->x_mode == SEP_PRO_NONE)))
            return 0;
        sep_wgrad_desc q = *d;
        q.G2 = reinterpret_cast<const float*>(d->G2_pre);      // same row pitch and line size as the fp32 tensor: the DMA does not change
        if (d->x_mode == SEP_PRO_PRELU) launch_w16<4, 1, SEP_PRO_PRELU, true>(q, stream);
        else launch_w16<4, 1, SEP_PRO_NONE, true>(q, stream);
        return 1;
    }
#define SEP_LW(XM)                                           \
    do {                                                     \
        launch_w16<4, 1, XM>(*d, stream);                    \
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

static bool w16_takes(const sep_wgrad_desc* d) {
    static const bool off = getenv("SEPK_WGRAD_F16") != nullptr && atoi(getenv("SEPK_WGRAD_F16")) == 0;
    if (off || d->arith != SEP_ARITH_F16X3 || d->g_mul || d->x_div != 1 || d->B > W16MAXB || d->g_split % 128 != 0) return false;
    if (d->M % 256 != 0 || d->N % 128 != 0) return false;
    if ((size_t)d->M * d->ldt * 4 >= (1ull << 32) || (size_t)d->N * d->ldt * 4 >= (1ull << 32)) return false;
    if ((long)d->nsplit > (long)d->B * (d->ldt / DK)) return false;
    return true;
}

// Called by sep_pw_wgrad_batch (gemm.hip): n products whose descriptors differ in G, G2, X, partial, partial_bias only (the caller has checked
// that).  Returns 1 when they were launched here as ONE grid.
int sep_pw_wgrad_pc16_batch(const sep_wgrad_desc* ds, int n, hipStream_t stream) {
    if (n < 1 || n > W16_MAXBATCH || !w16_takes(&ds[0])) return 0;
    W16Batch b;
    b.d = ds[0];
    const int ntiles = (ds[0].M / 256) * (ds[0].N / 128);
    b.per = 8 * ntiles * ceil_div(ds[0].nsplit, 8);
    for (int k = 0; k < W16_MAXBATCH; ++k) {
        const sep_wgrad_desc& q = ds[k < n ? k : 0];
        b.G[k] = q.G; b.G2[k] = q.G2; b.X[k] = q.X; b.partial[k] = q.partial; b.partial_bias[k] = q.partial_bias;
    }
    switch (ds[0].x_mode) {
        case SEP_PRO_NONE: launch_w16_batch<4, 1, SEP_PRO_NONE>(b, n, stream); break;
        case SEP_PRO_PRELU: launch_w16_batch<4, 1, SEP_PRO_PRELU>(b, n, stream); break;
        case SEP_PRO_GLN: launch_w16_batch<4, 1, SEP_PRO_GLN>(b, n, stream); break;
        default: launch_w16_batch<4, 1, SEP_PRO_GLN_PRELU>(b, n, stream); break;
    }
    return 1;
}
