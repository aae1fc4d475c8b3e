// Development tool: the single-instruction helpers of csrc/wgrad_pc16.hip on the device (the host simulation replaces them by C++):
// w16_max_halves (v_permlane32_swap), w16_max_neighbour (v_max_f32_dpp), w16_split2_pair (v_fma_mix_f32).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/w16_probe tools/w16_probe.hip && /tmp/w16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float w16_max_halves(const float m) {
    // v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second: the operands must be two REGISTERS (the
    // builtin, handed the same value twice, was given one register by hipcc and returned garbage: tools/w16_probe.hip), hence the copy in asm
    unsigned a = __builtin_bit_cast(unsigned, m), b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "=&v"(b));
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ float w16_max_neighbour(const float m) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(m));
    return r;
}
__global__ void probe(const float* in, float* o1, float* o2) {
    const float m = in[threadIdx.x];
    o1[threadIdx.x] = w16_max_halves(m);
    o2[threadIdx.x] = w16_max_neighbour(m);
}
int main() {
    float h[64], a[64], b[64], *d, *da, *db;
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64) + 0.5f;
    hipMalloc(&d, 256); hipMalloc(&da, 256); hipMalloc(&db, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, da, db);
    hipMemcpy(a, da, 256, hipMemcpyDeviceToHost); hipMemcpy(b, db, 256, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int i = 0; i < 64; ++i) {
        if (a[i] != fmaxf(h[i], h[i ^ 32])) ++bad1;
        if (b[i] != fmaxf(h[i], h[i ^ 1])) ++bad2;
    }
    printf("max_halves: %s (%d wrong)   max_neighbour: %s (%d wrong)\n", bad1 ? "FAIL" : "PASS", bad1, bad2 ? "FAIL" : "PASS", bad2);
    if (bad1) for (int i = 0; i < 64; i += 8) printf("  lane %2d: in %.1f partner %.1f got %.1f\n", i, h[i], h[i ^ 32], a[i]);
    return 0;
}
