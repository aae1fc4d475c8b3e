// Shared device/host helpers for libsepkernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sepkernels.h"

#define SEP_WAVE 64

// ---- error plumbing (thread-local, never throws across the ABI) -------------------
void sep_set_error(const char* fmt, ...);

#define SEP_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            sep_set_error(__VA_ARGS__);   \
            return -1;                    \
        }                                 \
    } while (0)

#define SEP_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            sep_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return -2;                                                               \
        }                                                                            \
    } while (0)

// ---- wave / block reductions --------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block (4 waves). `sm` must hold >= 4 elements of T. Result valid in thread 0
// (and in every thread of wave 0).  Ends with a barrier so `sm` can be reused.
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    T r = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    return r;
}

// Same for a workgroup of NW waves (NW <= 8).
template <typename T, int NW>
__device__ __forceinline__ T block_sum_n(T v, T* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    T r = sm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r += sm[i];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float prelu_f(float x, float a) { return x > 0.f ? x : a * x; }
__device__ __forceinline__ float prelu_grad(float x, float a) { return x > 0.f ? 1.f : a; }

// gLN statistics are accumulated with fp64 atomics into SEP_STATS_SLOTS independent {sum, sumsq} slots per sample
// (slot = blockIdx & 15) so that the ~10^4 producer blocks of one tensor do not serialise on two addresses.
#ifndef SEP_STATS_SLOTS
#define SEP_STATS_SLOTS 16
#endif

// mean / rstd of a gLN from its slots -- biased variance like nn.GroupNorm.  st points at the sample's first slot.
__device__ __forceinline__ void gln_mu_rstd_d(const double* st, double count, float eps, double& m, double& r) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < SEP_STATS_SLOTS; ++k) { s0 += st[2 * k]; s1 += st[2 * k + 1]; }
    m = s0 / count;
    double var = s1 / count - m * m;
    if (var < 0.0) var = 0.0;
    r = 1.0 / sqrt(var + (double)eps);
}
__device__ __forceinline__ void gln_mu_rstd(const double* st, double count, float eps, float& mu, float& rstd) {
    double m, r;
    gln_mu_rstd_d(st, count, eps, m, r);
    mu = (float)m;
    rstd = (float)r;
}
// gLN backward: the two per-sample means every input gradient needs,
//     mg = mean_{c,t}(gamma_c g),   mgx = mean_{c,t}(gamma_c g xhat),   xhat = (u - mu) rstd.
// Round 2 formed them in two second-stage launches between the kernel that produces g and the kernel that consumes it (98 launches per
// step on the critical path).  Now the PRODUCER finishes them: every workgroup adds its gamma-weighted row-sum totals
//     acc[slot] += { sum_c gamma_c sum_t g , sum_c gamma_c sum_t g u }                  (fp64 atomics, SEP_STATS_SLOTS slots like the statistics)
// and arrives at the sample's counters; the LAST arrival turns the slots into the two means and stores them where the consumer's
// prologue reads two floats (as it always did).  ONE thread per workgroup calls this with the workgroup's two totals.  Both sums are linear in the row sums, so they need neither mu nor rstd while they are accumulated.
// First form of this round: the CONSUMERS summed the slots -- 64 fp64 loads and an fp64 divide / sqrt in every workgroup of the
// depthwise backward (+8 us per launch) and spilled registers in the GEMM kernels.
// Arrival counters: SEP_STATS_SLOTS + 1 ints per sample.  A workgroup arrives at ITS SLOT's counter; the last of that slot's
// `expected_in_slot` arrivals arrives at the sample's counter [SEP_STATS_SLOTS]; the last of those `nslots` is the sample's last workgroup.
// Two levels because a returning atomic on ONE address costs about a microsecond and they serialise: 512 workgroups per sample on one
// counter made the depthwise backward take 504 us instead of 95 (profiles/r03e_kernel_stats.md); 32 per address disappear in its run time.
static_assert(SEP_ARRIVE_INTS == SEP_STATS_SLOTS + 1, "arrival counters");
// No fences: every access that has to be seen by another compute unit here is an agent-scope ATOMIC (performed at the level the XCDs
// share), and the thread waits for the values its own adds return before it arrives, so its sums are in place when its arrival is counted.
// A __threadfence() instead writes back / invalidates the whole L2 of the XCD on this part -- in a kernel streaming 134 MB through that
// L2 it took the depthwise backward from 95 to 504 us (profiles/r03e_kernel_stats.md, r03f_kernel_stats.md).
__device__ __forceinline__ void gln_bwd_publish(double* acc, const double* st, int* counters, float* means, int slot, int expected_in_slot,
                                                int nslots, double a1, double a2, double count, float eps) {
    const double o1 = atomicAdd(acc + 2 * slot, a1), o2 = atomicAdd(acc + 2 * slot + 1, a2);      // returning forms: waited for below
    asm volatile("" :: "v"(o1), "v"(o2));
    if (atomicAdd(counters + slot, 1) != expected_in_slot - 1) return;
    if (atomicAdd(counters + SEP_STATS_SLOTS, 1) != nslots - 1) return;
    double s1 = 0.0, s2 = 0.0, m, r;
#pragma unroll
    for (int k = 0; k < SEP_STATS_SLOTS; ++k) {        // coherent loads: the slots were written by atomics of other compute units
        s1 += __hip_atomic_load(acc + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s2 += __hip_atomic_load(acc + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gln_mu_rstd_d(st, count, eps, m, r);
    means[0] = (float)(s1 / count);
    means[1] = (float)(r * (s2 - m * s1) / count);
}
// Consumer-side form of the same means: from slots a producer only ADDED to (non-returning atomics, no arrival protocol).  For producers
// with thousands of short workgroups -- the depthwise backward: 8192 per launch -- the two waited-for round trips of gln_bwd_publish at
// the end of every workgroup cost more (+10 us per launch) than one thread of each consuming workgroup summing the 16 slots (+~1 us).
__device__ __forceinline__ void gln_bwd_means(const double* acc, const double* st, double count, float eps, float& mg, float& mgx) {
    double s1 = 0.0, s2 = 0.0, m, r;
#pragma unroll
    for (int k = 0; k < SEP_STATS_SLOTS; ++k) { s1 += acc[2 * k]; s2 += acc[2 * k + 1]; }
    gln_mu_rstd_d(st, count, eps, m, r);
    mg = (float)(s1 / count);
    mgx = (float)(r * (s2 - m * s1) / count);
}
// arrivals a slot sees when `n` workgroups numbered 0 .. n-1 arrive at slot (number & (SEP_STATS_SLOTS - 1)), and how many slots see any
__device__ __forceinline__ int arrivals_in_slot(int n, int slot) { return (n - slot + SEP_STATS_SLOTS - 1) / SEP_STATS_SLOTS; }
__device__ __forceinline__ int slots_in_use(int n) { return n < SEP_STATS_SLOTS ? n : SEP_STATS_SLOTS; }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// gemm_coop.hip: the packed-weight form of sep_pw_gemm (1 = launched, 0 = not one of its shapes)
int sep_pw_gemm_packed(const sep_gemm_desc* d, hipStream_t stream);
// gemm_pc.hip: its producer / consumer kernel (same contract)
int sep_pw_gemm_pc(const sep_gemm_desc* d, hipStream_t stream);
// wgrad_pc.hip: producer / consumer form of sep_pw_wgrad in the bf16 split arithmetic (same contract)
int sep_pw_wgrad_pc(const sep_wgrad_desc* d, hipStream_t stream);
// wgrad_pc16.hip: the same in the scaled two-part fp16 arithmetic (SEP_ARITH_F16X3; same contract)
int sep_pw_wgrad_pc16(const sep_wgrad_desc* d, hipStream_t stream);
int sep_pw_wgrad_pc16_batch(const sep_wgrad_desc* ds, int n, hipStream_t stream);
