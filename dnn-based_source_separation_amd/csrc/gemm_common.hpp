// Device helpers shared by the GEMM translation units (gemm.hip, gemm_coop.hip) of libsepkernels: float4 access, DPP row
// sums, the LDS-transposed float4 epilogue, LDS-DMA issue / wait helpers and the two-part fp16 split.  gfx950 only.
#pragma once
#include "common.hpp"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ const float* byte_off(const float* base, unsigned bytes) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)bytes);
}
// Sum over the 16 lanes of a DPP row (lanes 16i .. 16i+15), result in every lane of the row.  Four VALU with DPP operands
// (quad xor 1, quad xor 2, mirror inside 8, mirror inside 16: a sum does not care which partner it meets) instead of four
// ds_bpermute round trips through the LDS crossbar with an lgkmcnt wait each.
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
    return x;
}
// Slab store of the weight-gradient kernels.  (Round 2 prepared a second form that ADDED every slab onto one zeroed slab with the
// hardware fp32 atomic, so that the slabs would never exist in HBM; measured on MI355X in round 3 it is slower -- 100 vs 82 us per launch,
// 17.85 vs 17.29 ms per step, profiles/r03a_wgrad_atomic.txt: the atomics of 64 workgroups on one address resolve at the memory side
// across the eight XCDs -- and it is gone.)
template <bool ATOMIC>
__device__ __forceinline__ void wg_put(float* p, const float v) {
    static_assert(!ATOMIC, "the atomic slab accumulation was removed");
    *p = v;
}
// output tile store of the GEMM epilogue
__device__ __forceinline__ void st4_out(float* p, float4 v) {
    *reinterpret_cast<float4*>(p) = v;
}

// Shared epilogue of the GEMM kernels.  acc[mi][ni] are the wave's four 32x32 accumulators.
//
// The MFMA C layout gives a lane ONE column of 16 different rows, i.e. 4-byte stores.  Measured on the first
// version: 64 dword stores per lane made the epilogue 40 % of the short-K GEMMs (store-issue bound, ~2 TB/s).
// So each wave transposes its tile through LDS (32 rows at a time, wave-private region, conflict-free
// ds_write_b32 / ds_read_b128) and every global access of the epilogue -- result stores, residual / accumulate /
// aux reads -- is a float4 covering 256 contiguous bytes per 16 lanes.  All reads of a half tile are issued before
// its first store (a load placed after a store cannot be hoisted: possible alias).
constexpr int EPI_LD = 68;                         // floats per transposed row (64 + 4: keeps float4 alignment)
constexpr int EPI_WAVE_FLOATS = 32 * EPI_LD;       // LDS floats one wave needs

#ifdef SEP_PROF
__device__ long long g_prof[4][4][16];
__device__ long long g_blk_start[8192], g_blk_end[8192];      // [block sample][wave][stamp]
#define PROF_STAMP(k) do { if (prof_slot >= 0 && lane == 0) g_prof[prof_slot][wid][k] = clock64(); } while (0)
#else
#define PROF_STAMP(k) do { } while (0)
#endif
#ifdef SEP_PROF
#define PROF_ARG , const int prof_slot
#define PROF_PASS , prof_slot
#else
#define PROF_ARG
#define PROF_PASS
#endif
// EF >= 0: the epilogue flag set as a compile-time constant AND a promise of the host dispatch that M (and m_split) are
// multiples of 128, so no row predicate exists (the host instantiates this for the combinations the model uses);
// EF < 0: flags read from the descriptor, rows predicated.  Column edge (the last column tile of a sample, frames >= T):
// the (bias-added) tile is multiplied by a 0/1 lane mask in a small wave-uniform block BEFORE the flag-dependent math, so
// statistics, row sums and the stored pad frames come out as zeros without a second copy of the math.
// Why this is lean on purpose: when the other three waves of a SIMD are issuing MFMAs back to back, a VALU instruction of
// the epilogue wave gets an issue slot roughly once per MFMA (s_memtime stamps: the same epilogue took 14 k cycles alone,
// 47-58 k next to three busy waves, and it STRETCHED when the main loops were staggered away from it).  First version:
// run-time flags (~300 branches per tile), 64-bit address arithmetic and a predicate per row: 800-1600 VALU per
// tile-wave.  Now 150-800: addresses are a wave-uniform row pointer (SGPRs) plus one per-lane byte offset, the loads of a
// group are issued before the first use, flags and predicates fold away.  Keep it free of scratch: above ~200 B/lane the
// runtime falls back to per-dispatch scratch allocation (+25 us per launch, measured).
// MI: 32-row blocks per wave (the wave's tile is 32*MI rows x 64 columns); RS: the accumulators are in the scaled units of
// the packed-weight path and each output row is multiplied by d.a_rscale[row] (fused with the bias add); bn_tile: columns
// of the workgroup's tile (128, or 64 for the cooperative kernel whose waves are stacked along the rows: wc = 0).
// PIPE: the global reads of group g+1 (bias, row scale, residual / aux / accumulate operands) are issued before group g is
// processed (two operand buffers): one exposed memory round trip per call instead of one per group -- for kernels that run
// ONE workgroup per CU (pw_gemm_pc_kernel), where no other workgroup hides those round trips.
// NW: waves of the workgroup (the block reductions at the end are executed by ALL of them); active = false: this wave owns
// no output tile (the producer waves of pw_gemm_pc_kernel) and only takes part in those reductions.
template <int EF, int MI = 2, bool RS = false, int NW = 4, bool PIPE = false>
__device__ __forceinline__ void gemm_epilogue(const sep_gemm_desc& d, f32x16 (&acc)[MI][2], const int b, const int m0,
                                              const int t0, const int wr, const int wc, const int lk, const int l31,
                                              const int tid, float* lds, double* red, const int bn_tile,
                                              const bool active = true PROF_ARG) {
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ef = EF >= 0 ? EF : d.epi_flags;
    const float alpha_e = (ef & (SEP_EPI_STATS_PRELU | SEP_EPI_PRELU_BWD | SEP_EPI_ROWSUMS_PRELU)) ? d.epi_alpha[0] : 0.f;
    float st_s = 0.f, st_ss = 0.f, dalpha_e = 0.f;
    const int Mfirst = d.m_split ? d.m_split : d.M;
    const int wrow = m0 + wr * (32 * MI);                    // first output row of this wave
    const bool second = d.m_split && wrow >= d.m_split;      // wave-uniform: m_split is a multiple of 128, wrow of 32*MI
    const int Mdst = second ? d.M - d.m_split : Mfirst;
    const int rowoff = second ? d.m_split : 0;
    float* __restrict__ dst = second ? d.Y2 : d.Y;
    const bool acc_this = d.accumulate && (second || !d.m_split);
    const bool use_res = (ef & SEP_EPI_RESIDUAL) && !second;
    const bool use_aux = (ef & (SEP_EPI_PRELU_BWD | SEP_EPI_ROWSUMS)) != 0;
    const bool has_bias = d.bias != nullptr;
    float* Tw = lds + wid * EPI_WAVE_FLOATS;
    const int rsub = lane >> 4, c4 = lane & 15;              // read-back: 4 rows x 16 float4 per pass
    const int tc = t0 + wc * 64 + 4 * c4;
    // wave-uniform pointers to (first row of this wave, first frame of this tile); a lane adds lane_off bytes
    const unsigned lane_off = 4u * (unsigned)(rsub * d.ldt + wc * 64 + 4 * c4);
    float* const dst_w = dst + ((size_t)b * Mdst + (wrow - rowoff)) * d.ldt + t0;
    const float* const res_w = use_res ? d.epi_res + ((size_t)b * Mfirst + wrow) * d.ldt + t0 : nullptr;
    const float* const aux_w = use_aux ? d.epi_aux + ((size_t)b * d.M + wrow) * d.ldt + t0 : nullptr;
    const float* const bias_w = has_bias ? d.bias + wrow : nullptr;
    const float* const rs_w = RS ? d.a_rscale + wrow : nullptr;
    constexpr bool FULL = EF >= 0;          // rows never need a predicate
    constexpr int GRP = FULL ? 4 : 1;
    const bool full_cols = t0 + bn_tile <= d.T;  // block-uniform
    float cm[4];                            // 0/1 column mask of this lane's four frames
#pragma unroll
    for (int e = 0; e < 4; ++e) cm[e] = (tc + e) < d.T ? 1.f : 0.f;

    if (active) {
struct GrpOps { float4 ext[GRP], aux[GRP], old[GRP]; float bs[GRP], rs[GRP]; };
        // every global read of a group of GRP passes, issued back to back
        auto issue_loads = [&](const int mi, const int g4, GrpOps& o) {
            bool ok[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                ok[j] = FULL || (wrow + mi * 32 + (g4 + j) * 4 + rsub) < d.M;
                if (!FULL) {                       // rows past M: neutral operands (on full tiles every use is guarded by the same flag as its load)
                    o.bs[j] = 0.f;
                    o.ext[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    o.aux[j] = o.ext[j];
                    o.old[j] = o.ext[j];
                }
            }
            if (has_bias) {
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (ok[j]) o.bs[j] = *byte_off(bias_w + mi * 32 + (g4 + j) * 4, 4u * (unsigned)rsub);
            }
            if (RS) {
#pragma unroll
                for (int j = 0; j < GRP; ++j) o.rs[j] = ok[j] ? *byte_off(rs_w + mi * 32 + (g4 + j) * 4, 4u * (unsigned)rsub) : 0.f;
            }
            if (use_res) {
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (ok[j]) o.ext[j] = ld4(byte_off(res_w + (mi * 32 + (g4 + j) * 4) * (size_t)d.ldt, lane_off));
            }
            if (use_aux) {
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (ok[j]) o.aux[j] = ld4(byte_off(aux_w + (mi * 32 + (g4 + j) * 4) * (size_t)d.ldt, lane_off));
            }
            if (acc_this) {
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (ok[j]) o.old[j] = ld4(byte_off(dst_w + (mi * 32 + (g4 + j) * 4) * (size_t)d.ldt, lane_off));
            }
        };
        // transposed tile rows of the group -> bias / scale / flag-dependent math -> stores
        auto process = [&](const int mi, const int g4, GrpOps& o) {
            bool ok[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) ok[j] = FULL || (wrow + mi * 32 + (g4 + j) * 4 + rsub) < d.M;
            float4 outv[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) outv[j] = ld4(Tw + ((g4 + j) * 4 + rsub) * EPI_LD + 4 * c4);
            if (RS) {
                if (has_bias) {
#pragma unroll
                    for (int j = 0; j < GRP; ++j) {
                        outv[j].x = fmaf(outv[j].x, o.rs[j], o.bs[j]); outv[j].y = fmaf(outv[j].y, o.rs[j], o.bs[j]);
                        outv[j].z = fmaf(outv[j].z, o.rs[j], o.bs[j]); outv[j].w = fmaf(outv[j].w, o.rs[j], o.bs[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < GRP; ++j) { outv[j].x *= o.rs[j]; outv[j].y *= o.rs[j]; outv[j].z *= o.rs[j]; outv[j].w *= o.rs[j]; }
                }
            } else if (has_bias) {
#pragma unroll
                for (int j = 0; j < GRP; ++j) { outv[j].x += o.bs[j]; outv[j].y += o.bs[j]; outv[j].z += o.bs[j]; outv[j].w += o.bs[j]; }
            }
            if (!full_cols) {
#pragma unroll
                for (int j = 0; j < GRP; ++j) { outv[j].x *= cm[0]; outv[j].y *= cm[1]; outv[j].z *= cm[2]; outv[j].w *= cm[3]; }
            }
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                float v[4] = {outv[j].x, outv[j].y, outv[j].z, outv[j].w};
                float ax[4] = {0.f, 0.f, 0.f, 0.f};
                if (use_aux) { ax[0] = o.aux[j].x; ax[1] = o.aux[j].y; ax[2] = o.aux[j].z; ax[3] = o.aux[j].w; }
                float rs1 = 0.f, rs2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool live = FULL || ok[j];          // invalid frames already hold zeros
                    if (ef & SEP_EPI_STATS_PRELU) {
                        const float u = prelu_f(v[e], alpha_e);
                        if (live) { st_s += u; st_ss = fmaf(u, u, st_ss); }
                    }
                    // 1 / (1 + 2^(-v log2 e)) on the bare v_exp_f32 / v_rcp_f32 (1 ulp each; the libm forms cost ~17 VALU)
                    if (ef & SEP_EPI_SIGMOID) v[e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
                    if (ef & SEP_EPI_PRELU_BWD) {
                        if (live && ax[e] <= 0.f) dalpha_e = fmaf(v[e], ax[e], dalpha_e);
                        v[e] *= prelu_grad(ax[e], alpha_e);
                    }
                    if (ef & SEP_EPI_ROWSUMS) {
                        const float u = (ef & SEP_EPI_ROWSUMS_PRELU) ? prelu_f(ax[e], alpha_e) : ax[e];
                        if (live) { rs1 += v[e]; rs2 = fmaf(v[e], u, rs2); }
                    }
                    if ((ef & SEP_EPI_SIGMOID) && !full_cols) v[e] *= cm[e];     // sigmoid(0) = 0.5: mask again
                }
                outv[j] = make_float4(v[0], v[1], v[2], v[3]);
                if (ef & SEP_EPI_ROWSUMS) {
                    // the 16 lanes with equal (lane >> 4) share this row: xor offsets < 16 stay inside the group
                    rs1 = row16_sum(rs1);
                    rs2 = row16_sum(rs2);
                    if (c4 == 0 && ok[j]) {
                        float* rp = d.epi_rowpart + (((size_t)b * d.M + wrow + mi * 32 + (g4 + j) * 4 + rsub) * (d.ldt / 64) + (t0 + wc * 64) / 64) * 2;
                        rp[0] = rs1; rp[1] = rs2;
                    }
                }
            }
            if (use_res) {                  // whole-group block (use_res depends on which output part this tile is in)
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    // frames >= T of the residual / accumulated tensors are zero by contract, so the sums keep them zero
                    outv[j].x += o.ext[j].x; outv[j].y += o.ext[j].y; outv[j].z += o.ext[j].z; outv[j].w += o.ext[j].w;
                }
            }
            if (acc_this) {
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    outv[j].x += o.old[j].x; outv[j].y += o.old[j].y; outv[j].z += o.old[j].z; outv[j].w += o.old[j].w;
                }
            }
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                if (ok[j])
                    st4_out(const_cast<float*>(byte_off(dst_w + (mi * 32 + (g4 + j) * 4) * (size_t)d.ldt, lane_off)), outv[j]);
            }
        };
        constexpr int NGB = 8 / GRP;                  // groups per 32-row block
        GrpOps ops[PIPE ? 2 : 1];
        if (PIPE) issue_loads(0, 0, ops[0]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        // ---- transpose: registers -> LDS (C layout) --------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * lk;
            Tw[rl * EPI_LD + l31] = acc[mi][0][r];
            Tw[rl * EPI_LD + 32 + l31] = acc[mi][1][r];
        }
        __builtin_amdgcn_wave_barrier();       // LDS is in-order per wave; this only pins the compiler's order
#ifdef SEP_PROF
        __builtin_amdgcn_s_waitcnt(0x0070);
        PROF_STAMP(8 + 4 * mi);
#endif
        // ---- per group of GRP passes (4 rows each): every global read issued back to back, then compute + stores.
        //      Four on full tiles (eight would spill: 3 x 8 float4 of operands next to the second half's accumulators),
        //      two on edge tiles, whose predicates need registers too -- ANY scratch in this kernel costs occupancy.
        //      Wave-uniform options (bias / residual / accumulate) are whole-group blocks: one scalar branch each.
#pragma unroll
        for (int gi = 0; gi < NGB; ++gi) {
            const int s = mi * NGB + gi;
            if (PIPE) {
                // the next group's operands travel while this group is computed and stored (next group may be in the next block)
                if (s + 1 < MI * NGB) issue_loads((s + 1) / NGB, ((s + 1) % NGB) * GRP, ops[(s + 1) & 1]);
                process(mi, gi * GRP, ops[s & 1]);
            } else {
                issue_loads(mi, gi * GRP, ops[0]);
                process(mi, gi * GRP, ops[0]);
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef SEP_PROF
        PROF_STAMP(10 + 4 * mi);
#endif
    }
    }   // active
    if (ef & SEP_EPI_STATS_PRELU) {
        const double s = block_sum_n<double, NW>((double)st_s, red);
        const double ss = block_sum_n<double, NW>((double)st_ss, red);
        if (tid == 0) { double* st = d.epi_stats + ((size_t)b * SEP_STATS_SLOTS + (blockIdx.x & (SEP_STATS_SLOTS - 1))) * 2; atomicAdd(st, s); atomicAdd(st + 1, ss); }
    }
    if (ef & SEP_EPI_PRELU_BWD) {
        const double s = block_sum_n<double, NW>((double)dalpha_e, red);
        if (tid == 0) atomicAdd(d.epi_dalpha, s);
    }
}

// ======================================================================================
// Direct-to-LDS staging helpers shared by the fast GEMM / wgrad kernels.
// global_load_lds_dwordx4: the 64 lanes of a wave copy 64 x 16 B from per-lane global addresses to ONE contiguous
// 1 KiB LDS range (wave-uniform base in M0 + lane*16) without touching VGPRs.
// ======================================================================================
constexpr int DK = 16;      // contraction rows per ring stage
constexpr int NST = 4;      // ring depth of the weight-gradient kernel

// s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier as BUILTINS: the compiler's waitcnt pass then knows every counter is zero
// here and emits counted lgkmcnt(N) waits afterwards (behind an opaque asm it falls back to lgkmcnt(0) everywhere)
// all but the newest `keep4` DMA instructions of this wave have landed (keep4 in {0, 4}), then the barrier
__device__ __forceinline__ void wait_keep4_and_barrier(const bool keep4) {
    asm volatile("" ::: "memory");
    if (keep4) __builtin_amdgcn_s_waitcnt(0x0074);
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// all but the DMAs of the newest `chunks` chunks (4 instructions each) of this wave have landed, then the barrier
__device__ __forceinline__ void wait_chunks_and_barrier(const int chunks) {
    asm volatile("" ::: "memory");
    if (chunks >= 3) __builtin_amdgcn_s_waitcnt(0x007c);
    else if (chunks == 2) __builtin_amdgcn_s_waitcnt(0x0078);
    else if (chunks == 1) __builtin_amdgcn_s_waitcnt(0x0074);
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_all_and_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// LDS-DMA of 16 B per lane as an asm statement, saddr form: source = base (SGPR pair) + zext(voff), LDS image lane-linear
// from the wave-uniform byte address lds_dst.  Why not the builtin: hipcc books a global_load_lds as a FLAT access that may
// touch LDS, and from then on every LDS-read dependency in the loop becomes s_waitcnt lgkmcnt(0) -- no counted waits, so
// a ds_read could never stay in flight across an MFMA burst.  Behind asm the DMA is invisible to that bookkeeping (its
// completion is waited for by hand: vmcnt(0) before the barrier that publishes the stage).  M0 is compiler-reserved:
// saved and restored in the same statement.
__device__ __forceinline__ void glds16_asm(const float* base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// the same for data a kernel reads exactly once (the activation stream of the forward / input-gradient products): with -DSEPK_DMA_NT the
// request carries the non-temporal hint, so the stream does not push the packed weights every workgroup re-reads out of the XCD's L2
__device__ __forceinline__ void glds16_asm_once(const float* base, unsigned voff, unsigned lds_dst) {
#ifdef SEPK_DMA_NT
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
#else
    glds16_asm(base, voff, lds_dst);
#endif
}
// same, with a full 64-bit per-lane source address
__device__ __forceinline__ void glds16_asm_v(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)reinterpret_cast<size_t>((const __attribute__((address_space(3))) float*)p);
}
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
// AR == 2 (SEP_ARITH_F16X3): fp32 products from a TWO-part fp16 split, x*2^s = hi + lo with hi = fp16(x*2^s) (toward zero),
// lo = fp16(x*2^s - hi) (11 + 11 significand bits), three part products hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16: 3 x 32 matrix-pipe cycles and 6 VALU per pair of values where the bf16 split needs 6 x 32 and 11.
// fp16 has 5 exponent bits, so the operands are brought into range with exact power-of-two scales: ONE for A, from a
// caller-supplied upper bound of |A| (largest scaled value < 2^13), and for B one PER COLUMN, kept per lane (a column of
// the B tile is a lane of the MFMA operand and owns its accumulator column) and lowered on the fly: when a chunk's column
// maximum would pass 2^14 the lane's accumulators are rescaled (rare after the first chunks); both are undone on the
// accumulators before the epilogue.  Values more than ~2^-25 below their column's maximum lose low bits (fp16 underflow):
// the error is relative to |A||X| per output like fp32 accumulation's, not elementwise -- tools/split_accuracy.py,
// tools/gemm_accuracy.py and the kernel tests put it at the fp32-MFMA path's level on operands spread over e^+-6.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(const float x0, const float x1, unsigned& hi, unsigned& lo) {
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const fp16x2_t l = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h.x, x1 - (float)h.y);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split2_frag(const float (&lo4)[4], const float (&hi4)[4], u32x4_t (&out)[2]) {
    unsigned p[2][4];
    split2_pair(lo4[0], lo4[1], p[0][0], p[1][0]);
    split2_pair(lo4[2], lo4[3], p[0][1], p[1][1]);
    split2_pair(hi4[0], hi4[1], p[0][2], p[1][2]);
    split2_pair(hi4[2], hi4[3], p[0][3], p[1][3]);
#pragma unroll
    for (int q = 0; q < 2; ++q) out[q] = u32x4_t{p[q][0], p[q][1], p[q][2], p[q][3]};
}
__device__ __forceinline__ void mfma_split3(const u32x4_t (&a)[2], const u32x4_t (&b)[2], f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[0]), __builtin_bit_cast(f16x8_t, b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[1]), __builtin_bit_cast(f16x8_t, b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[0]), __builtin_bit_cast(f16x8_t, b[0]), acc, 0, 0, 0);
}

}  // namespace
