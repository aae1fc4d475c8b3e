// Layer norm over the FEATURES of token-major rows, with the residual sum in front of it (ABI 22).
//
// The post-norm transformer layers of SepFormer (reference src/models/sepformer.py:395-520: nn.TransformerEncoderLayer, `norm1(x + dropout1(sa(x)))`,
// `norm2(x + dropout2(ff(x)))`) and GALRNet's channel norm in front of its attention (src/models/galr.py:172-190 LayerNormAlongChannel) normalise
// every token's C features on their own -- nn.LayerNorm(C): mean and biased variance over the row, eps inside the root, gain / shift per feature.
// As torch kernels that is five passes per site (dropout, add, norm; norm's input gradient, its parameter gradients, the dropout's mask), the
// row read or written eleven times; here one pass each way:
//   forward : s = x + drop(r) ; y = (s - mu) rstd gamma + beta ; stat[row] = {mu, rstd}                         reads x, r ; writes s, y
//   backward: ds = rstd (g gamma - mean(g gamma) - xhat mean(g gamma xhat)) ; dr = drop'(ds)                    reads dy, s ; writes ds [, dr]
//             part[w] = {sum_rows dy xhat [C] | sum_rows dy [C]} of the rows workgroup w took
// drop() is the inverted dropout of nn.Dropout with a mask that is a FUNCTION of (seed, element index) -- the hash of csrc/attn.hip -- so the
// backward pass forms it again instead of reading a mask tensor.  One wave per row (64 lanes x float4 x NJ trips, C <= 1024), rows strided over
// the grid's waves; mean and variance in two passes over the REGISTERS (no E[x^2] - E[x]^2).  HBM-bound: 4 rows of C floats each way.
#include "common.hpp"

namespace {

__device__ __forceinline__ float4 rn_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void rn_st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }

// the mixing of att_hash (csrc/attn.hip): 64-bit element index, 64-bit seed as two words
__device__ __forceinline__ unsigned rn_hash(const unsigned long long idx, const unsigned s0, const unsigned s1) {
    unsigned x = (unsigned)idx ^ s0;
    x *= 0x9E3779B1u; x ^= x >> 15;
    x *= 0x85EBCA6Bu; x ^= x >> 13;
    x += s1 + (unsigned)(idx >> 32) * 0x9E3779B1u;
    x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

struct rn_drop {
    unsigned thr, s0, s1;       // thr == 0: no dropout
    float keep_inv;
};

rn_drop rn_make_drop(const float p_drop, const unsigned long long seed) {
    rn_drop d;
    double t = (double)p_drop * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    d.thr = p_drop > 0.f ? (unsigned)t : 0u;
    if (p_drop > 0.f && d.thr == 0u) d.thr = 1u;
    d.keep_inv = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    d.s0 = (unsigned)(seed & 0xffffffffull);
    d.s1 = (unsigned)(seed >> 32);
    return d;
}

template <int NJ>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ s_out, float* __restrict__ y,
                                                          float* __restrict__ stat, long rows, int C, float eps, rn_drop dr) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 g4[NJ], b4[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 4 * lane + 256 * j;
        g4[j] = c < C ? rn_ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        b4[j] = c < C ? rn_ld4(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_c = 1.f / (float)C;
    for (long row = (long)blockIdx.x * 4 + w; row < rows; row += (long)gridDim.x * 4) {
        const size_t base = (size_t)row * C;
        float v[NJ][4];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 4 * lane + 256 * j;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                a = rn_ld4(x + base + c);
                if (r) {
                    const float4 q = rn_ld4(r + base + c);
                    float qv[4] = {q.x, q.y, q.z, q.w};
                    if (dr.thr != 0u) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) qv[e] = rn_hash((unsigned long long)(base + c + e), dr.s0, dr.s1) >= dr.thr ? qv[e] * dr.keep_inv : 0.f;
                    }
                    a.x += qv[0]; a.y += qv[1]; a.z += qv[2]; a.w += qv[3];
                    if (s_out) rn_st4(s_out + base + c, a);
                }
            }
            v[j][0] = a.x; v[j][1] = a.y; v[j][2] = a.z; v[j][3] = a.w;
            sum += (a.x + a.y) + (a.z + a.w);
        }
        const float mu = wave_sum(sum) * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (4 * lane + 256 * j < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; sq = fmaf(d, d, sq); }
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(sq) * inv_c + eps);
        if (lane == 0) { stat[2 * row] = mu; stat[2 * row + 1] = rstd; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 4 * lane + 256 * j;
            if (c < C)
                rn_st4(y + base + c, make_float4(fmaf((v[j][0] - mu) * rstd, g4[j].x, b4[j].x), fmaf((v[j][1] - mu) * rstd, g4[j].y, b4[j].y),
                                                 fmaf((v[j][2] - mu) * rstd, g4[j].z, b4[j].z), fmaf((v[j][3] - mu) * rstd, g4[j].w, b4[j].w)));
        }
    }
}

template <int NJ>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s, const float* __restrict__ gamma,
                                                          const float* __restrict__ stat, float* __restrict__ ds, float* __restrict__ dres,
                                                          float* __restrict__ part, long rows, int C, rn_drop dr) {
    __shared__ float pc[4][2][256 * NJ];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ga[NJ][4], pg[NJ][4], pb[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 4 * lane + 256 * j;
        const float4 g4 = c < C ? rn_ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        ga[j][0] = g4.x; ga[j][1] = g4.y; ga[j][2] = g4.z; ga[j][3] = g4.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pg[j][e] = 0.f; pb[j][e] = 0.f; }
    }
    const float inv_c = 1.f / (float)C;
    for (long row = (long)blockIdx.x * 4 + w; row < rows; row += (long)gridDim.x * 4) {
        const size_t base = (size_t)row * C;
        const float mu = stat[2 * row], rstd = stat[2 * row + 1];
        float gg[NJ][4], xh[NJ][4];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 4 * lane + 256 * j;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f), v = make_float4(mu, mu, mu, mu);
            if (c < C) { g = rn_ld4(dy + base + c); v = rn_ld4(s + base + c); }
            const float gv[4] = {g.x, g.y, g.z, g.w}, sv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[j][e] = (sv[e] - mu) * rstd;
                gg[j][e] = gv[e] * ga[j][e];
                a1 += gg[j][e];
                a2 = fmaf(gg[j][e], xh[j][e], a2);
                pg[j][e] = fmaf(gv[e], xh[j][e], pg[j][e]);
                pb[j][e] += gv[e];
            }
        }
        const float m1 = wave_sum(a1) * inv_c, m2 = wave_sum(a2) * inv_c;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = 4 * lane + 256 * j;
            if (c < C) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (gg[j][e] - m1 - xh[j][e] * m2);
                rn_st4(ds + base + c, make_float4(o[0], o[1], o[2], o[3]));
                if (dres) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rn_hash((unsigned long long)(base + c + e), dr.s0, dr.s1) >= dr.thr ? o[e] * dr.keep_inv : 0.f;
                    rn_st4(dres + base + c, make_float4(o[0], o[1], o[2], o[3]));
                }
            }
        }
    }
    // the four waves' per-feature sums meet in LDS; thread t adds up feature t (+ 256 j)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pc[w][0][256 * j + 4 * lane + e] = pg[j][e];
            pc[w][1][256 * j + 4 * lane + e] = pb[j][e];
        }
    __syncthreads();
    float* ps = part + (size_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        ps[c] = (pc[0][0][c] + pc[1][0][c]) + (pc[2][0][c] + pc[3][0][c]);
        ps[C + c] = (pc[0][1][c] + pc[1][1][c]) + (pc[2][1][c] + pc[3][1][c]);
    }
}

// ReLU followed by the inverted dropout of the feed-forward sub-block (`dropout(activation(linear1(x)))`, torch/nn/modules/transformer.py
// _ff_block): a = keep ? max(h, 0) / (1 - p) : 0, mask from the same hash; the backward needs no mask and no h -- a is zero exactly where the
// gradient is: dh = a != 0 ? dy / (1 - p) : 0 (a is saved by the product that consumes it anyway).  Two streams forward, three backward.
__global__ __launch_bounds__(256) void relu_drop_fwd_kernel(const float* __restrict__ h, float* __restrict__ a, long n4, rn_drop dr) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = rn_ld4(h + 4 * i);
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool keep = dr.thr == 0u || rn_hash((unsigned long long)(4 * i + e), dr.s0, dr.s1) >= dr.thr;
            o[e] = (o[e] > 0.f && keep) ? o[e] * dr.keep_inv : 0.f;
        }
        rn_st4(a + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a, float* __restrict__ dh, long n4,
                                                            float keep_inv) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 g = rn_ld4(dy + 4 * i), v = rn_ld4(a + 4 * i);
        rn_st4(dh + 4 * i, make_float4(v.x != 0.f ? g.x * keep_inv : 0.f, v.y != 0.f ? g.y * keep_inv : 0.f, v.z != 0.f ? g.z * keep_inv : 0.f,
                                       v.w != 0.f ? g.w * keep_inv : 0.f));
    }
}

int rn_flat_grid(long n4) {
    const long want = (n4 + 255) / 256;
    return (int)(want < 256 * 16 ? want : 256 * 16);
}

bool rn_shape_ok(long rows, int C) { return rows > 0 && C >= 4 && C <= 1024 && C % 4 == 0; }

}  // namespace

/* workgroups sep_rownorm_* launch at this shape = slabs of `part` the backward writes: eight workgroups of four waves per compute unit while
 * the rows last */
extern "C" int sep_rownorm_parts(long rows, int C) {
    if (!rn_shape_ok(rows, C)) return 0;
    const long want = (rows + 3) / 4;
    return (int)(want < 2048 ? want : 2048);
}

extern "C" int sep_rownorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* s, float* y, float* stat, long rows,
                               int C, float eps, float p_drop, unsigned long long seed, sep_stream_t stream) {
    SEP_REQUIRE(x && gamma && beta && y && stat && rn_shape_ok(rows, C), "sep_rownorm_fwd: bad arguments (C a multiple of 4, at most 1024)");
    SEP_REQUIRE(res != nullptr || s == nullptr, "sep_rownorm_fwd: a place for the sum without a residual branch");      // (s may be NULL with a branch: inference)
    SEP_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || res), "sep_rownorm_fwd: 0 <= p_drop < 1, and only on a residual branch");
    const rn_drop d = rn_make_drop(p_drop, seed);
    const int grid = sep_rownorm_parts(rows, C), nj = (C + 255) / 256;
#define SEP_RNF(NJ) hipLaunchKernelGGL((rownorm_fwd_kernel<NJ>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, s, y, stat, rows, C, eps, d)
    if (nj == 1) SEP_RNF(1); else if (nj == 2) SEP_RNF(2); else if (nj == 3) SEP_RNF(3); else SEP_RNF(4);
#undef SEP_RNF
    SEP_CHECK_LAUNCH("sep_rownorm_fwd");
    return 0;
}

extern "C" int sep_rownorm_bwd(const float* dy, const float* s, const float* gamma, const float* stat, float* ds, float* dres, float* part, long rows,
                               int C, float p_drop, unsigned long long seed, sep_stream_t stream) {
    SEP_REQUIRE(dy && s && gamma && stat && ds && part && rn_shape_ok(rows, C), "sep_rownorm_bwd: bad arguments (C a multiple of 4, at most 1024)");
    SEP_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop > 0.f) == (dres != nullptr), "sep_rownorm_bwd: dres exactly when the branch was dropped out");
    const rn_drop d = rn_make_drop(p_drop, seed);
    const int grid = sep_rownorm_parts(rows, C), nj = (C + 255) / 256;
#define SEP_RNB(NJ) hipLaunchKernelGGL((rownorm_bwd_kernel<NJ>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, s, gamma, stat, ds, dres, part, rows, C, d)
    if (nj == 1) SEP_RNB(1); else if (nj == 2) SEP_RNB(2); else if (nj == 3) SEP_RNB(3); else SEP_RNB(4);
#undef SEP_RNB
    SEP_CHECK_LAUNCH("sep_rownorm_bwd");
    return 0;
}

extern "C" int sep_relu_drop_fwd(const float* h, float* a, long n, float p_drop, unsigned long long seed, sep_stream_t stream) {
    SEP_REQUIRE(h && a && n > 0 && n % 4 == 0 && p_drop >= 0.f && p_drop < 1.f, "sep_relu_drop_fwd: bad arguments (n a multiple of 4, 0 <= p_drop < 1)");
    hipLaunchKernelGGL(relu_drop_fwd_kernel, dim3(rn_flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, h, a, n / 4, rn_make_drop(p_drop, seed));
    SEP_CHECK_LAUNCH("sep_relu_drop_fwd");
    return 0;
}

extern "C" int sep_relu_drop_bwd(const float* dy, const float* a, float* dh, long n, float p_drop, sep_stream_t stream) {
    SEP_REQUIRE(dy && a && dh && n > 0 && n % 4 == 0 && p_drop >= 0.f && p_drop < 1.f, "sep_relu_drop_bwd: bad arguments (n a multiple of 4, 0 <= p_drop < 1)");
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3(rn_flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, a, dh, n / 4, rn_make_drop(p_drop, 0ull).keep_inv);
    SEP_CHECK_LAUNCH("sep_relu_drop_bwd");
    return 0;
}
