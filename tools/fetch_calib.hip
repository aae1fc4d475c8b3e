// Development tool: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access patterns
// the kernels of this library use.  MI355X_MICROARCH.md states that FETCH_SIZE reports half the bytes of a wide coalesced read (16 B per
// lane, 128-byte requests tallied at 64) and that "other access widths and WRITE_SIZE are uncalibrated" -- tools/pmc_traffic.py doubles
// FETCH_SIZE for every kernel, which is only right if every kernel's reads look like pattern 0 below.  Each kernel moves exactly
// `bytes` (printed) of a buffer far larger than the 256 MiB Infinity Cache, touching every byte once:
//   read_contig16     a wave reads 1 KiB contiguous per instruction (float4 per lane): the streaming kernels (depthwise, decoder ...)
//   read_rows64       a wave reads 16 rows x 64 B per instruction (lane >> 2 = row, lane & 3 = 16-byte piece), walking along the
//                     rows 64 B at a time: the weight-gradient kernels' operand fetch (global_load_dwordx4 form)
//   read_rows64_dma   the same addresses through global_load_lds_dwordx4 (what pw_wgrad_pc_kernel issues)
//   read_rows64_dma_spaced   the same with ~1 us between the two 64-byte halves of a line (the weight-gradient kernel's real timing)
//   read_rows128_dma  a wave reads 8 rows x 128 B per instruction through the LDS DMA: pw_gemm_pc_kernel's operand fetch
//   write_contig16    a wave writes 1 KiB contiguous per instruction (the transposed GEMM epilogue, the streaming kernels)
//   write_rows4       a lane writes 4 B, 32 lanes = 128 B of one row, two rows per instruction: the gLN-backward store-back
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip
//     rocprofv3 --pmc FETCH_SIZE -d /tmp/fc_f -- /tmp/fetch_calib ; rocprofv3 --pmc WRITE_SIZE -d /tmp/fc_w -- /tmp/fetch_calib
//     python tools/pmc_summary.py /tmp/fc_f "%"   (KiB per dispatch; compare with the printed byte counts)
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int ROWS = 8192, LDT = 16384;                  // 8192 rows x 64 KiB = 512 MiB
constexpr size_t N = (size_t)ROWS * LDT;

__global__ __launch_bounds__(256) void read_contig16(const float4* __restrict__ x, float* out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
// workgroup = 4 waves = 64 rows; blockIdx.y = column slab of 4096 floats
__global__ __launch_bounds__(256) void read_rows64(const float* __restrict__ x, float* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + wv * 16 + (lane >> 2);
    const float* p = x + (size_t)row * LDT + blockIdx.y * 4096 + 4 * (lane & 3);
    float s = 0.f;
    for (int c = 0; c < 4096 / 16; ++c) { const float4 v = *reinterpret_cast<const float4*>(p + 16 * c); s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__global__ __launch_bounds__(256) void read_rows64_dma(const float* __restrict__ x, float* out) {
    __shared__ __attribute__((aligned(16))) float buf[4][4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + wv * 16 + (lane >> 2);
    const float* p = x + (size_t)row * LDT + blockIdx.y * 4096 + 4 * (lane & 3);
    float s = 0.f;
    for (int c = 0; c < 4096 / 16; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) glds16(p + 16 * (c + u), &buf[wv][u][0]);
        __builtin_amdgcn_s_waitcnt(0x0070);
        s += buf[wv][c & 3][lane];
    }
    if (s == 12345.678f) out[0] = s;
}
// read_rows64_dma with the two 64-byte halves of every 128-byte line requested ~a microsecond apart (a dependent ALU chain between the
// chunks), as in pw_wgrad_pc_kernel where the second half belongs to the NEXT chunk: do the halves still merge into one 128-byte request?
__global__ __launch_bounds__(256) void read_rows64_dma_spaced(const float* __restrict__ x, float* out) {
    __shared__ __attribute__((aligned(16))) float buf[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 64 + wv * 16 + (lane >> 2);
    const float* p = x + (size_t)row * LDT + blockIdx.y * 4096 + 4 * (lane & 3);
    float s = 0.f;
    for (int c = 0; c < 4096 / 16; ++c) {
        glds16(p + 16 * c, &buf[wv][0]);
        __builtin_amdgcn_s_waitcnt(0x0070);
        float t = buf[wv][lane];
#pragma unroll 1
        for (int k = 0; k < 400; ++k) t = t * 1.0001f + 0.5f;        // ~400 dependent VALU ~ 1 us
        s += t;
    }
    if (s == 12345.678f) out[0] = s;
}
// a wave reads 8 rows x 128 B per instruction; workgroup = 4 waves = 32 rows... walking 128 B at a time
__global__ __launch_bounds__(256) void read_rows128_dma(const float* __restrict__ x, float* out) {
    __shared__ __attribute__((aligned(16))) float buf[4][4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 32 + wv * 8 + (lane >> 3);
    const float* p = x + (size_t)row * LDT + blockIdx.y * 4096 + 4 * (lane & 7);
    float s = 0.f;
    for (int c = 0; c < 4096 / 32; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) glds16(p + 32 * (c + u), &buf[wv][u][0]);
        __builtin_amdgcn_s_waitcnt(0x0070);
        s += buf[wv][c & 3][lane];
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void write_contig16(float4* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) y[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// wave: lanes 0-31 -> 32 consecutive floats of row r, lanes 32-63 -> of row r + 8 (the C layout of the producers' store-back); 16 rows per
// wave-iteration, workgroup = 64 rows x a 4096-float slab
__global__ __launch_bounds__(256) void write_rows4(float* __restrict__ y) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* p = y + (size_t)(blockIdx.x * 64 + wv * 16 + 8 * (lane >> 5)) * LDT + blockIdx.y * 4096 + (lane & 31);
    for (int c = 0; c < 4096 / 32; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) p[(size_t)e * LDT + 32 * c] = (float)(c + e);
}

int main() {
    float *x, *y, *out;
    hipMalloc(&x, N * 4); hipMalloc(&y, N * 4); hipMalloc(&out, 4);
    hipMemset(x, 0, N * 4); hipMemset(y, 0, N * 4);
    printf("every kernel moves %zu bytes = %.1f KiB = %.1f MB\n", N * 4, N * 4 / 1024.0, N * 4 / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read_contig16, dim3(8192), dim3(256), 0, 0, (const float4*)x, out, N / 4);
        hipLaunchKernelGGL(read_rows64, dim3(ROWS / 64, LDT / 4096), dim3(256), 0, 0, x, out);
        hipLaunchKernelGGL(read_rows64_dma, dim3(ROWS / 64, LDT / 4096), dim3(256), 0, 0, x, out);
        hipLaunchKernelGGL(read_rows128_dma, dim3(ROWS / 32, LDT / 4096), dim3(256), 0, 0, x, out);
        hipLaunchKernelGGL(read_rows64_dma_spaced, dim3(ROWS / 64, LDT / 4096), dim3(256), 0, 0, x, out);
        hipLaunchKernelGGL(write_contig16, dim3(8192), dim3(256), 0, 0, (float4*)y, N / 4);
        hipLaunchKernelGGL(write_rows4, dim3(ROWS / 64, LDT / 4096), dim3(256), 0, 0, y);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("done: %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
