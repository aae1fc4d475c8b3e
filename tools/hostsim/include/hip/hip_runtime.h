// Development / test tool (tools/hostsim.py): the pieces of the HIP programming model that the asm-free kernel files of csrc/ use
// (stream.hip, cln.hip, loss.hip, lstm.hip), restated for the HOST, so that the kernels' own source -- compiled as plain C++ with this
// directory in front of the include path -- runs as ordinary threads: one thread per lane of a workgroup, a pthread barrier per workgroup
// and per wave, wave shuffles and the MFMA instructions as collective operations of a wave.  Not a general HIP emulator.  Known limits:
// collectives must be reached by every lane of the wave / workgroup (true of these kernels; divergent shuffles would deadlock here),
// `__shared__` is a function-local static (one workgroup runs at a time), no streams (launches complete before they return).
#pragma once
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "host simulation"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
template <typename T> static inline T min(T a, T b) { return b < a ? b : a; }
template <typename T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static

// A barrier for many more threads than cores: arrivals are counted, waiters yield until the generation changes (pthread barriers put every
// waiter to sleep on a futex and wake them one by one: several times slower at 256 - 512 threads per workgroup).
struct SimBarrier {
    int n = 1;
    int count = 0;
    int generation = 0;
    void init(int threads) { n = threads; count = 0; generation = 0; }
    void wait() {
        const int g = __atomic_load_n(&generation, __ATOMIC_ACQUIRE);
        if (__atomic_add_fetch(&count, 1, __ATOMIC_ACQ_REL) == n) {
            __atomic_store_n(&count, 0, __ATOMIC_RELAXED);
            __atomic_store_n(&generation, g + 1, __ATOMIC_RELEASE);
        } else {
            int spins = 0;
            while (__atomic_load_n(&generation, __ATOMIC_ACQUIRE) == g) { if (++spins > 64) sched_yield(); }
        }
    }
};

struct SimCtx {
    std::vector<unsigned long long> dynamic_lds;      // `extern __shared__ T name[]` -- tools/hostsim.py rewrites that one declaration form
    SimBarrier block_barrier;
    std::vector<SimBarrier> wave_barrier;
    std::vector<unsigned long long> slot, slot2;      // exchange slots, one (pair) per thread of the workgroup
    std::vector<unsigned long long> wide, wide2;      // 16-byte slots (the packed MFMA operands)
};
extern SimCtx* g_sim;
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
#define warpSize 64
static inline void* sim_dynamic_lds() { return g_sim->dynamic_lds.data(); }

static inline void __syncthreads() { g_sim->block_barrier.wait(); }
static inline void sim_wave_sync() { g_sim->wave_barrier[threadIdx.x >> 6].wait(); }

template <typename T>
static inline T sim_read_lane(T v, int src_lane) {                     // every lane of the wave publishes v, then reads lane src_lane's
    static_assert(sizeof(T) <= 8, "shuffle of a type wider than 8 bytes");
    const unsigned t = threadIdx.x, base = t & ~63u;
    unsigned long long w = 0;
    memcpy(&w, &v, sizeof(T));
    g_sim->slot[t] = w;
    sim_wave_sync();
    const unsigned src = base + (unsigned)(src_lane & 63);
    const unsigned long long r = src < g_sim->slot.size() ? g_sim->slot[src] : w;
    sim_wave_sync();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) { const int l = threadIdx.x & 63; return sim_read_lane(v, (l & ~(width - 1)) + (src & (width - 1))); }
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { const int l = threadIdx.x & 63; (void)width; return sim_read_lane(v, l ^ mask); }
template <typename T> static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    const int l = threadIdx.x & 63, in_seg = l & (width - 1);
    const T got = sim_read_lane(v, in_seg >= (int)delta ? l - (int)delta : l);
    return got;
}
template <typename T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    const int l = threadIdx.x & 63, in_seg = l & (width - 1);
    return sim_read_lane(v, in_seg + (int)delta < width ? l + (int)delta : l);
}

static inline float atomicAdd(float* p, float v) {
    uint32_t* q = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(q, __ATOMIC_RELAXED), want;
    float f;
    do { memcpy(&f, &old, 4); f += v; memcpy(&want, &f, 4); } while (!__atomic_compare_exchange_n(q, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}
static inline double atomicAdd(double* p, double v) {
    uint64_t* q = reinterpret_cast<uint64_t*>(p);
    uint64_t old = __atomic_load_n(q, __ATOMIC_RELAXED), want;
    double f;
    do { memcpy(&f, &old, 8); f += v; memcpy(&want, &f, 8); } while (!__atomic_compare_exchange_n(q, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 8);
    return f;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __HIP_MEMORY_SCOPE_AGENT 4
template <typename T> static inline T sim_atomic_load(const T* p, int order) { T v; __atomic_load(const_cast<T*>(p), &v, order); return v; }
template <typename T, typename V> static inline void sim_atomic_store(T* p, V v, int order) { T t = (T)v; __atomic_store(p, &t, order); }
#define __hip_atomic_load(p, order, scope) sim_atomic_load(p, order)
#define __hip_atomic_store(p, v, order, scope) sim_atomic_store(p, v, order)
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }          // the hardware fp32 atomic of the device build
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }          // only used on wave-uniform values
static inline float __builtin_amdgcn_rcpf(float x) { return 1.f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.f / sqrtf(x); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { sim_wave_sync(); }        // lanes are threads here: the hardware's lock-step is a barrier
static inline float __fdividef(float a, float b) { return a / b; }
static inline unsigned __float_as_uint(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
static inline float __uint_as_float(unsigned u) { float x; memcpy(&x, &u, 4); return x; }
static inline int __float_as_int(float x) { int u; memcpy(&u, &x, 4); return u; }
static inline float __int_as_float(int u) { float x; memcpy(&x, &u, 4); return x; }

typedef float sim_f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_4x4x1_16b_f32: D_l[v] = C_l[v] + A_{4 (l / 4) + v} * B_l
static inline sim_f32x4 __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, sim_f32x4 c, int, int, int) {
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    memcpy(&g_sim->slot[t], &a, 4); memcpy(&g_sim->slot2[t], &b, 4);
    sim_wave_sync();
    sim_f32x4 d = c;
    float bv; memcpy(&bv, &g_sim->slot2[t], 4);
    for (int v = 0; v < 4; ++v) { float av; memcpy(&av, &g_sim->slot[base + 4 * (l / 4) + v], 4); d[v] += av * bv; }
    sim_wave_sync();
    return d;
}
// v_mfma_f32_16x16x4_f32: A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16], D[i = 4 (l / 16) + v][j = l % 16]
static inline sim_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, sim_f32x4 c, int, int, int) {
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    memcpy(&g_sim->slot[t], &a, 4); memcpy(&g_sim->slot2[t], &b, 4);
    sim_wave_sync();
    sim_f32x4 d = c;
    for (int v = 0; v < 4; ++v)
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, &g_sim->slot[base + 16 * k + 4 * (l / 16) + v], 4); memcpy(&bv, &g_sim->slot2[base + 16 * k + (l % 16)], 4);
            d[v] += av * bv;
        }
    sim_wave_sync();
    return d;
}

// one host thread per lane; the threads walk the grid together, one workgroup at a time
template <typename K, typename... A>
static void sim_launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
    const unsigned nt = block.x * block.y * block.z, nw = (nt + 63) / 64;
    SimCtx ctx;
    ctx.dynamic_lds.assign(shmem / 8 + 2, 0);
    ctx.block_barrier.init(nt);
    ctx.wave_barrier.resize(nw);
    for (unsigned w = 0; w < nw; ++w) ctx.wave_barrier[w].init((w + 1) * 64 <= nt ? 64 : nt - w * 64);
    ctx.slot.assign(nt, 0); ctx.slot2.assign(nt, 0);
    ctx.wide.assign(2 * nt, 0); ctx.wide2.assign(2 * nt, 0);
    g_sim = &ctx;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([=]() {
            blockDim = block; gridDim = grid;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = dim3(bx, by, bz);
                        kernel(args...);
                        g_sim->block_barrier.wait();        // the workgroup's statics (its LDS) are free for the next one
                    }
        });
    for (auto& x : th) x.join();
    g_sim = nullptr;
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) sim_launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)

// =====================================================================================================================================
// The GEMM files (gemm.hip, gemm_coop.hip, gemm_pc.hip, wgrad_pc.hip) use more of the machine: LDS-DMA, DPP, byte permutes, packed
// conversions, the 32x32 MFMA shapes, hand-placed waits.  tools/hostsim.py rewrites a few helper bodies in the copies it compiles
// (the inline-assembly ones); everything else is restated here.  LDS-DMA lands immediately (the hardware's asynchrony is not modelled:
// waits are no-ops), LDS byte addresses are offsets from an anchor object of the library.
#include <sched.h>
extern char g_lds_anchor;
static inline unsigned sim_lds_addr(const void* p) { return (unsigned)(int)((const char*)p - &g_lds_anchor); }
static inline char* sim_lds_ptr(unsigned a) { return &g_lds_anchor + (int)a; }
static inline void sim_glds16(const void* src_lane, unsigned lds_dst) {       // one DMA instruction of a wave: no lane is ahead of or behind it
    sim_wave_sync();
    memcpy(sim_lds_ptr(lds_dst) + 16 * (threadIdx.x & 63u), src_lane, 16);
    sim_wave_sync();
}

#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
static inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
static inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }
static inline int __builtin_amdgcn_frexp_expf(float x) { int e = 0; if (x != 0.f && isfinite(x)) frexpf(x, &e); return e; }
static inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long src = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned b = (sel >> (8 * i)) & 0xffu;
        const unsigned byte = b <= 7 ? (unsigned)((src >> (8 * b)) & 0xffu) : (b >= 0x0du ? 0xffu : 0u);
        r |= byte << (8 * i);
    }
    return r;
}
static inline void __builtin_amdgcn_global_load_lds(const void* g, void* lds_wave_base, int size, int, int) {
    sim_wave_sync();
    memcpy((char*)lds_wave_base + size * (threadIdx.x & 63u), g, size);
    sim_wave_sync();
}

typedef __fp16 sim_fp16x2 __attribute__((ext_vector_type(2)));
static inline uint16_t sim_f2h_rtz(float f) {                                  // float -> half bits, round toward zero
    _Float16 h = (_Float16)f;                                                   // nearest-even first
    const float back = (float)h;
    uint16_t bits; memcpy(&bits, &h, 2);
    if (f == f && fabsf(back) > fabsf(f)) bits -= 1;                            // one step back toward zero (also inf -> 65504)
    return bits;
}
static inline sim_fp16x2 __builtin_amdgcn_cvt_pkrtz(float a, float b) {
    const uint16_t h[2] = {sim_f2h_rtz(a), sim_f2h_rtz(b)};
    sim_fp16x2 r; memcpy(&r, h, 4);
    return r;
}

static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
    const int l = threadIdx.x & 63;
    int from;
    if (ctrl < 0x100) from = (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3);           // quad_perm
    else if (ctrl == 0x140) from = (l & ~15) + 15 - (l & 15);                    // row_mirror
    else if (ctrl == 0x141) from = (l & ~7) + 7 - (l & 7);                       // row_half_mirror
    else { fprintf(stderr, "hostsim: DPP control 0x%x is not modelled\n", ctrl); abort(); }
    (void)old;
    return sim_read_lane(src, from);
}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) {
    const unsigned t = threadIdx.x, base = t & ~63u;
    g_sim->slot[t] = p ? 1ull : 0ull;
    sim_wave_sync();
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64 && base + i < g_sim->slot.size(); ++i) m |= g_sim->slot[base + i] << i;
    sim_wave_sync();
    return m;
}

typedef float sim_f32x16 __attribute__((ext_vector_type(16)));
// 32 x 32 accumulator layout: register r of lane l is D[(r % 4) + 8 (r / 4) + 4 (l / 32)][l % 32]
static inline sim_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, sim_f32x16 c, int, int, int) {   // A[i = l % 32][k = l / 32], B[k][j = l % 32]
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    memcpy(&g_sim->slot[t], &a, 4); memcpy(&g_sim->slot2[t], &b, 4);
    sim_wave_sync();
    sim_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const unsigned i = (r % 4) + 8 * (r / 4) + 4 * (l / 32), j = l % 32;
        for (unsigned k = 0; k < 2; ++k) { float av, bv; memcpy(&av, &g_sim->slot[base + i + 32 * k], 4); memcpy(&bv, &g_sim->slot2[base + j + 32 * k], 4); d[r] += av * bv; }
    }
    sim_wave_sync();
    return d;
}
template <typename V, typename ToFloat>
static inline sim_f32x16 sim_mfma_32x32x16(V a, V b, sim_f32x16 c, ToFloat tof) {           // lane l holds A[i = l % 32][8 (l / 32) + 0..7], B likewise
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    memcpy(&g_sim->wide[2 * t], &a, 16); memcpy(&g_sim->wide2[2 * t], &b, 16);
    sim_wave_sync();
    sim_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const unsigned i = (r % 4) + 8 * (r / 4) + 4 * (l / 32), j = l % 32;
        float s = 0.f;
        for (unsigned k = 0; k < 16; ++k) {
            V av, bv;
            memcpy(&av, &g_sim->wide[2 * (base + i + 32 * (k / 8))], 16); memcpy(&bv, &g_sim->wide2[2 * (base + j + 32 * (k / 8))], 16);
            s += tof(av, k % 8) * tof(bv, k % 8);
        }
        d[r] += s;
    }
    sim_wave_sync();
    return d;
}
typedef _Float16 sim_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 sim_bf16x8;
static inline sim_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(sim_f16x8 a, sim_f16x8 b, sim_f32x16 c, int, int, int) {
    return sim_mfma_32x32x16(a, b, c, [](const sim_f16x8& v, unsigned e) { return (float)v[e]; });
}
static inline sim_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(sim_bf16x8 a, sim_bf16x8 b, sim_f32x16 c, int, int, int) {
    return sim_mfma_32x32x16(a, b, c, [](const sim_bf16x8& v, unsigned e) { uint16_t h; memcpy(&h, (const char*)&v + 2 * e, 2); uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; });
}
