// Development tool: the fp32 MFMA speed of light actually reachable on the box (sustained clock included).
// A register-only loop of v_mfma_f32_32x32x2_f32 (4 independent accumulators per wave), W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/mfma_peak.hip && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int iters : {2000, 20000}) {
            const int blocks = 256 * wps;   // 4 waves per block = one per SIMD; wps blocks per CU
            mfma_loop<<<blocks, 256>>>(out, clk, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            mfma_loop<<<blocks, 256>>>(out, clk, iters);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
            printf("waves/SIMD %d iters %6d: %8.3f ms  %7.1f TF/s   shader clock %.0f MHz (s_memtime %lld ticks / wall %lld x 10 ns)\n",
                   wps, iters, ms, flop / ms / 1e9, (double)h[0] / ((double)h[1] * 0.01), h[0], h[1]);
        }
    }
    return 0;
}
