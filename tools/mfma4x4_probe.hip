// Development tool: the operand layout of v_mfma_f32_4x4x1_16b_f32 that csrc/lstm.hip's four-sequence sweeps (lstm_fwd4_kernel /
// lstm_bwd4_kernel, SEPK_LSTM_NS4) are written against, checked on the device with one-hot operands:
//   lane l supplies A[block l / 4][row l % 4] and B[block l / 4][column l % 4]; register v of lane l is D[block l / 4][row v][column l % 4]
// i.e. D_l[v] = A_{4 (l / 4) + v} * B_l for one instruction on a zero accumulator.  Prints PASS or the first mismatch, then the cycle
// cost of a dependent chain and of four independent chains (the sweeps use four).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma4x4_probe tools/mfma4x4_probe.hip && gpurun_out/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void one(const float* a, const float* b, float* d) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[threadIdx.x * 4 + v] = acc[v];
}

__global__ void chains(float* out, long long* clk, int iters) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    const float x = 1e-3f * threadIdx.x, y = 1.f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 32; ++k) a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
    long long t1 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, a3, 0, 0, 0);
        }
    long long t2 = clock64();
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = t2 - t1; }
}

int main() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    long long hc[2], *c;
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&d, sizeof(hd)); hipMalloc(&c, sizeof(hc));
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }      // distinct values: every product identifies its two lanes
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64 && !bad; ++l)
        for (int v = 0; v < 4; ++v) {
            const float want = ha[4 * (l / 4) + v] * hb[l];
            if (hd[4 * l + v] != want) {
                printf("MISMATCH lane %d register %d: got %g, the assumed layout gives %g (A of lane %d times B of lane %d)\n", l, v, hd[4 * l + v], want, 4 * (l / 4) + v, l);
                for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (ha[la] * hb[lb] == hd[4 * l + v]) printf("  it is A of lane %d times B of lane %d\n", la, lb);
                bad = 1; break;
            }
        }
    printf(bad ? "layout: FAIL\n" : "layout: PASS (D_l[v] = A_{4 (l/4) + v} * B_l)\n");
    hipLaunchKernelGGL(chains, dim3(1), dim3(64), 0, 0, d, c, 1000);
    hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    printf("32000 MFMAs, one wave: one dependent chain %.1f cycles per instruction, four chains %.1f\n", hc[0] / 32000.0, hc[1] / 32000.0);
    return bad;
}
