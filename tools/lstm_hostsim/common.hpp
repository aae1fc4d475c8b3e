// Development tool (tools/lstm_hostsim.py): the pieces of the HIP programming model csrc/lstm.hip uses, restated for the HOST so that the
// kernels' own source runs as ordinary threads -- one thread per lane, a pthread barrier per workgroup and per wave, and
// v_mfma_f32_4x4x1_16b_f32 / v_mfma_f32_16x16x4_f32 as collective operations of a wave under the operand layouts the kernels assume.
// Not a general HIP emulator: exactly what lstm.hip needs.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

typedef void* sep_stream_t;
typedef void* hipStream_t;
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

#define __global__
#define __device__
#define __forceinline__ inline
#define __launch_bounds__(n)
#define __restrict__
#define __shared__ static          /* one workgroup runs at a time: one static instance IS the workgroup's LDS */

struct SimCtx {
    pthread_barrier_t block_barrier;
    std::vector<pthread_barrier_t> wave_barrier;
    std::vector<float> xa, xb;     // MFMA operand exchange, one slot per thread of the workgroup
};
extern SimCtx* g_sim;
extern thread_local dim3 threadIdx, blockIdx;

static inline void __syncthreads() { pthread_barrier_wait(&g_sim->block_barrier); }
static inline void sim_wave_sync() { pthread_barrier_wait(&g_sim->wave_barrier[threadIdx.x >> 6]); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }          // only used on wave-uniform values
static inline float __builtin_amdgcn_rcpf(float x) { return 1.f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { sim_wave_sync(); }        // lanes are threads here: the hardware's lock-step is a barrier

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_4x4x1_16b_f32: D_l[v] = C_l[v] + A_{4 (l / 4) + v} * B_l
static inline f32x4 __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, f32x4 c, int, int, int) {
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    g_sim->xa[t] = a; g_sim->xb[t] = b;
    sim_wave_sync();
    f32x4 d = c;
    for (int v = 0; v < 4; ++v) d[v] += g_sim->xa[base + 4 * (l / 4) + v] * g_sim->xb[t];
    sim_wave_sync();
    return d;
}
// v_mfma_f32_16x16x4_f32: A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16], D[i = 4 (l / 16) + v][j = l % 16]
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    g_sim->xa[t] = a; g_sim->xb[t] = b;
    sim_wave_sync();
    f32x4 d = c;
    for (int v = 0; v < 4; ++v)
        for (int k = 0; k < 4; ++k) d[v] += g_sim->xa[base + 16 * k + 4 * (l / 16) + v] * g_sim->xb[base + 16 * k + (l % 16)];
    sim_wave_sync();
    return d;
}

void sep_set_error(const char* fmt, ...);
#define SEP_REQUIRE(cond, ...) do { if (!(cond)) { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return -1; } } while (0)
#define SEP_CHECK_LAUNCH(name) ((void)0)

template <typename K, typename... A>
static void sim_launch(K kernel, dim3 grid, dim3 block, A... args) {
    const unsigned nt = block.x, nw = (nt + 63) / 64;
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            SimCtx ctx;
            pthread_barrier_init(&ctx.block_barrier, nullptr, nt);
            ctx.wave_barrier.resize(nw);
            for (unsigned w = 0; w < nw; ++w) pthread_barrier_init(&ctx.wave_barrier[w], nullptr, (w + 1) * 64 <= nt ? 64 : nt - w * 64);
            ctx.xa.assign(nt, 0.f); ctx.xb.assign(nt, 0.f);
            g_sim = &ctx;
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([=]() { threadIdx = dim3(t); blockIdx = dim3(bx, by); kernel(args...); });
            for (auto& x : th) x.join();
            g_sim = nullptr;
        }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) sim_launch(kernel, grid, block, __VA_ARGS__)
