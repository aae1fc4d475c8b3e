// SI-SDR / PIT / Sinkhorn-PIT and the fused clip+Adam update for gfx950.
//
// Reference arithmetic replaced (under /root/reference/src): criterion/sdr.py:122-139 (sisdr),
// criterion/pit.py:9-44 (pit: loop over n! permutations), criterion/pit.py:163-193 (sinkpit);
// egs/wsj0-mix/common/src/driver.py:152-155 (clip_grad_norm_ + Adam.step).
//
// The O(T) work is two kernels: one pass producing the three dot products per (est_i, tgt_j) pair in
// fp64 (wavefront shuffles + one fp64 atomic per block), and one elementwise pass applying the analytic
// gradient  d_est_i = sum_j gw_ij * (cT_ij * tgt_j + cE_ij * est_i).  Everything in between (n x n
// matrices, n! search, Sinkhorn iterations) is O(n^2) per utterance and runs in tiny kernels.
#include "common.hpp"

namespace {

constexpr int DOT_CHUNK = 256 * 16;

// grid: (nchunk, npairs, B); pair p -> (i, j) = all_pairs ? (p / n, p % n) : (p, p)
__global__ __launch_bounds__(256) void sisdr_dots_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                         double* __restrict__ dots, double* __restrict__ tt,
                                                         double* __restrict__ xx, int n, int T, int all_pairs) {
    __shared__ double red[4];
    const int b = blockIdx.z, p = blockIdx.y;
    const int i = all_pairs ? p / n : p, j = all_pairs ? p % n : p;
    const float* e = est + ((size_t)b * n + i) * T;
    const float* t = tgt + ((size_t)b * n + j) * T;
    const bool do_tt = all_pairs ? (i == 0) : true;
    const bool do_xx = all_pairs ? (j == 0) : true;
    const int beg = blockIdx.x * DOT_CHUNK;
    float s_et = 0.f, s_tt = 0.f, s_xx = 0.f;
    double d_et = 0.0, d_tt = 0.0, d_xx = 0.0;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int idx = beg + k * 256 + threadIdx.x;
        if (idx < T) {
            const float ev = e[idx], tv = t[idx];
            s_et = fmaf(ev, tv, s_et); s_tt = fmaf(tv, tv, s_tt); s_xx = fmaf(ev, ev, s_xx);
        }
        if ((k & 3) == 3) {   // short fp32 runs, fp64 across them
            d_et += (double)s_et; d_tt += (double)s_tt; d_xx += (double)s_xx;
            s_et = s_tt = s_xx = 0.f;
        }
    }
    const double r_et = block_sum_256<double>(d_et, red);
    const double r_tt = block_sum_256<double>(d_tt, red);
    const double r_xx = block_sum_256<double>(d_xx, red);
    if (threadIdx.x == 0) {
        atomicAdd(dots + ((size_t)b * n + i) * n + j, r_et);
        if (do_tt) atomicAdd(tt + (size_t)b * n + j, r_tt);
        if (do_xx) atomicAdd(xx + (size_t)b * n + i, r_xx);
    }
}

struct SdrTerms { double alpha, c, S, Nn; };
__device__ __forceinline__ SdrTerms sdr_terms(double a, double ttv, double xxv, double eps) {
    SdrTerms r;
    r.c = ttv + eps;
    r.alpha = a / r.c;
    r.S = r.alpha * r.alpha * ttv + eps;
    double nn = r.alpha * r.alpha * ttv - 2.0 * r.alpha * a + xxv;   // |alpha t - x|^2
    if (nn < 0.0) nn = 0.0;
    r.Nn = nn + eps;
    return r;
}

__global__ void sisdr_from_dots_kernel(const double* __restrict__ dots, const double* __restrict__ tt,
                                       const double* __restrict__ xx, float* __restrict__ out, int B, int n,
                                       int all_pairs, float eps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n * n) return;
    const int b = idx / (n * n), i = (idx / n) % n, j = idx % n;
    if (!all_pairs && i != j) { out[idx] = 0.f; return; }
    const SdrTerms r = sdr_terms(dots[idx], tt[b * n + j], xx[b * n + i], (double)eps);
    out[idx] = (float)(10.0 * log10(r.S / r.Nn));
}

// grid: (ceil(T/1024), n, B); block 256 threads x 4 elements
__global__ __launch_bounds__(256) void sisdr_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                        const double* __restrict__ dots, const double* __restrict__ tt,
                                                        const double* __restrict__ xx, const float* __restrict__ gw,
                                                        float* __restrict__ d_est, int n, int T, int all_pairs, float eps) {
    __shared__ float cT[64];
    __shared__ float cE;
    const int b = blockIdx.z, i = blockIdx.y;
    if (threadIdx.x < 64) cT[threadIdx.x] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double K = 10.0 / log(10.0);
        double ce = 0.0;
        for (int j = 0; j < n; ++j) {
            if (!all_pairs && j != i) continue;
            const double g = (double)gw[((size_t)b * n + i) * n + j];
            if (g == 0.0) continue;
            const double a = dots[((size_t)b * n + i) * n + j], ttv = tt[b * n + j], xxv = xx[b * n + i];
            const SdrTerms r = sdr_terms(a, ttv, xxv, (double)eps);
            // d sisdr/dx = K [ (2 alpha tt / c) t / S - ( ((2 alpha tt - 2a)/c - 2 alpha) t + 2 x ) / N ]
            const double ct = K * (2.0 * r.alpha * ttv / (r.c * r.S) - ((2.0 * r.alpha * ttv - 2.0 * a) / r.c - 2.0 * r.alpha) / r.Nn);
            cT[j] = (float)(g * ct);
            ce += g * K * (-2.0 / r.Nn);
        }
        cE = (float)ce;
    }
    __syncthreads();
    const float* e = est + ((size_t)b * n + i) * T;
    float* o = d_est + ((size_t)b * n + i) * T;
    const float cev = cE;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (t < T) {
            float v = cev * e[t];
            for (int j = 0; j < n; ++j) {
                const float c = cT[j];
                if (c != 0.f) v = fmaf(c, tgt[((size_t)b * n + j) * T + t], v);
            }
            o[t] = v;
        }
    }
}

// one thread per batch item; permutations in itertools order, first extremum wins (torch.min/max semantics)
__global__ void pit_search_kernel(const float* __restrict__ val, const int32_t* __restrict__ perms, int P, int n, int B,
                                  int maximize, int use_mean, float* __restrict__ best_val, int64_t* __restrict__ best_idx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* v = val + (size_t)b * n * n;
    float best = 0.f;
    int bi = 0;
    for (int p = 0; p < P; ++p) {
        float s = 0.f;
        for (int k = 0; k < n; ++k) s += v[k * n + perms[p * n + k]];
        if (use_mean) s /= (float)n;
        if (p == 0 || (maximize ? (s > best) : (s < best))) { best = s; bi = p; }
    }
    best_val[b] = best;
    best_idx[b] = bi;
}

// ---- Sinkhorn: one 64-thread block per batch item, all iterates kept for the reverse sweep --------
__device__ __forceinline__ void lse_step(double* Z, double* lse, int n, int over_rows, int tid) {
    // over_rows = 1: logsumexp over the first index i for every j (torch dim=1 of (B,n,n)); else over j for every i
    for (int q = tid; q < n; q += 64) {
        double mx = -1e300;
        for (int r = 0; r < n; ++r) { const double z = over_rows ? Z[r * n + q] : Z[q * n + r]; mx = z > mx ? z : mx; }
        double s = 0.0;
        for (int r = 0; r < n; ++r) { const double z = over_rows ? Z[r * n + q] : Z[q * n + r]; s += exp(z - mx); }
        lse[q] = mx + log(s);
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 64) Z[e] -= over_rows ? lse[e % n] : lse[e / n];
    __syncthreads();
}

__global__ __launch_bounds__(64) void sinkhorn_fwd_kernel(const float* __restrict__ C, double* __restrict__ zwork,
                                                          float* __restrict__ loss, float* __restrict__ Pout, int n,
                                                          float coldness, int iters) {
    extern __shared__ __attribute__((aligned(16))) double sh[];   // Z[n*n], lse[n], red[64]
    double* Z = sh;
    double* lse = sh + n * n;
    double* red = lse + n;
    const int b = blockIdx.x, tid = threadIdx.x, nn = n * n;
    const float* Cb = C + (size_t)b * nn;
    double* zw = zwork + (size_t)b * (2 * iters + 1) * nn;
    const double beta = (double)coldness;
    for (int e = tid; e < nn; e += 64) { Z[e] = -beta * (double)Cb[e]; zw[e] = Z[e]; }
    __syncthreads();
    for (int h = 1; h <= 2 * iters; ++h) {
        lse_step(Z, lse, n, h & 1, tid);
        for (int e = tid; e < nn; e += 64) zw[(size_t)h * nn + e] = Z[e];
    }
    double acc = 0.0;
    for (int e = tid; e < nn; e += 64) {
        const double p = exp(Z[e]);
        Pout[(size_t)b * nn + e] = (float)p;
        acc += ((double)Cb[e] + Z[e] / beta) * p;
    }
    red[tid] = acc;
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int k = 0; k < 64; ++k) s += red[k]; loss[b] = (float)s; }
}

__global__ __launch_bounds__(64) void sinkhorn_bwd_kernel(const float* __restrict__ C, const double* __restrict__ zwork,
                                                          const float* __restrict__ dloss, float* __restrict__ dC, int n,
                                                          float coldness, int iters) {
    extern __shared__ __attribute__((aligned(16))) double sh[];   // dZ[n*n], red[n]
    double* dZ = sh;
    double* red = sh + n * n;
    const int b = blockIdx.x, tid = threadIdx.x, nn = n * n;
    const float* Cb = C + (size_t)b * nn;
    const double* zw = zwork + (size_t)b * (2 * iters + 1) * nn;
    const double beta = (double)coldness, g = (double)dloss[b];
    const double* Zf = zw + (size_t)(2 * iters) * nn;
    for (int e = tid; e < nn; e += 64) {
        const double p = exp(Zf[e]);
        dZ[e] = g * p * (1.0 / beta + (double)Cb[e] + Zf[e] / beta);
    }
    __syncthreads();
    for (int h = 2 * iters; h >= 1; --h) {
        const int over_rows = h & 1;
        const double* Zh = zw + (size_t)h * nn;        // Z_h = Z_{h-1} - LSE  ->  softmax(Z_{h-1}) = exp(Z_h)
        for (int q = tid; q < n; q += 64) {
            double s = 0.0;
            for (int r = 0; r < n; ++r) s += over_rows ? dZ[r * n + q] : dZ[q * n + r];
            red[q] = s;
        }
        __syncthreads();
        for (int e = tid; e < nn; e += 64) dZ[e] -= exp(Zh[e]) * (over_rows ? red[e % n] : red[e / n]);
        __syncthreads();
    }
    for (int e = tid; e < nn; e += 64) {
        const double p = exp(Zf[e]);
        dC[(size_t)b * nn + e] = (float)(g * p - beta * dZ[e]);
    }
}

// ---- row difference sums: the O(T) part of the distance criteria and plain SDR ---------------------------
// One workgroup per row (rows = every leading index of the reduced axis).  sums[row] = {sum |x-t|, sum (x-t)^2,
// sum t^2}: fp32 partials per thread over <= T/256 elements, fp64 across the workgroup.
template <bool VEC>
__global__ __launch_bounds__(256) void rowdiff_sums_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                           double* __restrict__ sums, int T) {
    __shared__ double red[4];
    const int64_t base = (int64_t)blockIdx.x * T;
    float sa = 0.f, sq = 0.f, st = 0.f;
    if (VEC) {
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        const float4* t4 = reinterpret_cast<const float4*>(t + base);
        for (int i = threadIdx.x; i < T / 4; i += 256) {
            const float4 a = x4[i], b = t4[i];
            const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
            sa += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
            sq = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, sq))));
            st = fmaf(b.x, b.x, fmaf(b.y, b.y, fmaf(b.z, b.z, fmaf(b.w, b.w, st))));
        }
    } else {
        for (int i = threadIdx.x; i < T; i += 256) {
            const float b = t[base + i], d = x[base + i] - b;
            sa += fabsf(d);
            sq = fmaf(d, d, sq);
            st = fmaf(b, b, st);
        }
    }
    const double ra = block_sum_256<double>((double)sa, red);
    const double rq = block_sum_256<double>((double)sq, red);
    const double rt = block_sum_256<double>((double)st, red);
    if (threadIdx.x == 0) {
        sums[3 * (int64_t)blockIdx.x + 0] = ra;
        sums[3 * (int64_t)blockIdx.x + 1] = rq;
        sums[3 * (int64_t)blockIdx.x + 2] = rt;
    }
}

// dx[row][i] = c_abs[row] * sign(x - t) + c_sq[row] * (x - t); grid = (column chunks of 1024, rows).
template <bool VEC>
__global__ __launch_bounds__(256) void rowdiff_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                          const float* __restrict__ c_abs, const float* __restrict__ c_sq,
                                                          float* __restrict__ dx, int T) {
    const int64_t base = (int64_t)blockIdx.y * T;
    const float ca = c_abs ? c_abs[blockIdx.y] : 0.f, cs = c_sq ? c_sq[blockIdx.y] : 0.f;
    auto g = [&](float d) { return fmaf(cs, d, d > 0.f ? ca : (d < 0.f ? -ca : 0.f)); };
    if (VEC) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < T / 4) {
            const float4 a = reinterpret_cast<const float4*>(x + base)[i], b = reinterpret_cast<const float4*>(t + base)[i];
            reinterpret_cast<float4*>(dx + base)[i] = make_float4(g(a.x - b.x), g(a.y - b.y), g(a.z - b.z), g(a.w - b.w));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = blockIdx.x * 1024 + k * 256 + threadIdx.x;
            if (i < T) dx[base + i] = g(x[base + i] - t[base + i]);
        }
    }
}

// ---- clip + Adam on a flat fp32 buffer ----------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, double* __restrict__ out, int64_t n) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t idx = i + k * 256;
            if (idx < n) { const float v = g[idx]; s = fmaf(v, v, s); }
        }
        acc += (double)s;
    }
    const double r = block_sum_256<double>(acc, red);
    if (threadIdx.x == 0) atomicAdd(out, r);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const double* __restrict__ sqnorm, int64_t n,
                                                   float lr, float b1, float b2, float eps, float wd, float max_norm,
                                                   float grad_scale, float bc1, float bc2s) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float coef = grad_scale;
    if (max_norm > 0.f) {
        const float total = grad_scale * (float)sqrt(sqnorm[0]);
        const float c = max_norm / (total + 1e-6f);
        coef *= (c < 1.f ? c : 1.f);
    }
    float gi = g[i] * coef;
    g[i] = gi;                       // clip_grad_norm_ rescales .grad in place
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

// Graph-replayable form: the step count and the learning rate live in device memory (a captured launch freezes its kernel
// arguments), the count is advanced by a one-thread kernel in front of this one.
__global__ void adam_tick_kernel(int* step) { step[0] += 1; }

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, const double* __restrict__ sqnorm, int64_t n,
                                                       const float* __restrict__ lr_dev, const int* __restrict__ step_dev, float b1, float b2,
                                                       float eps, float wd, float max_norm, float grad_scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = lr_dev[0];
    const float st = (float)step_dev[0];
    const float bc1 = 1.f - powf(b1, st);
    const float bc2s = sqrtf(1.f - powf(b2, st));
    float coef = grad_scale;
    if (max_norm > 0.f) {
        const float total = grad_scale * (float)sqrt(sqnorm[0]);
        const float c = max_norm / (total + 1e-6f);
        coef *= (c < 1.f ? c : 1.f);
    }
    float gi = g[i] * coef;
    g[i] = gi;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

}  // namespace

extern "C" int sep_adam_step_dev(float* p, float* g, float* m, float* v, const double* sqnorm, int64_t n, const float* lr_dev,
                                 int32_t* step_dev, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                                 float grad_scale, sep_stream_t stream) {
    SEP_REQUIRE(p && g && m && v && n > 0 && lr_dev && step_dev, "sep_adam_step_dev: bad arguments");
    SEP_REQUIRE(max_norm <= 0.f || sqnorm, "sep_adam_step_dev: clipping needs sqnorm");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sqnorm, n, lr_dev, step_dev,
                       beta1, beta2, eps, weight_decay, max_norm, grad_scale);
    SEP_CHECK_LAUNCH("sep_adam_step_dev");
    return 0;
}

extern "C" int sep_sisdr_dots(const float* est, const float* tgt, double* dots, double* tt, double* xx, int B, int n, int T,
                              int all_pairs, sep_stream_t stream) {
    SEP_REQUIRE(est && tgt && dots && tt && xx && B > 0 && n > 0 && T > 0, "sep_sisdr_dots: bad arguments");
    SEP_REQUIRE(n <= 64 && B <= 65535, "sep_sisdr_dots: n <= 64 and B <= 65535 supported");
    dim3 grid(ceil_div(T, DOT_CHUNK), all_pairs ? n * n : n, B);
    hipLaunchKernelGGL(sisdr_dots_kernel, grid, dim3(256), 0, (hipStream_t)stream, est, tgt, dots, tt, xx, n, T, all_pairs);
    SEP_CHECK_LAUNCH("sep_sisdr_dots");
    return 0;
}

extern "C" int sep_sisdr_from_dots(const double* dots, const double* tt, const double* xx, float* sisdr, int B, int n,
                                   int all_pairs, float eps, sep_stream_t stream) {
    SEP_REQUIRE(dots && tt && xx && sisdr && B > 0 && n > 0, "sep_sisdr_from_dots: bad arguments");
    hipLaunchKernelGGL(sisdr_from_dots_kernel, dim3(ceil_div(B * n * n, 256)), dim3(256), 0, (hipStream_t)stream, dots, tt, xx, sisdr, B, n, all_pairs, eps);
    SEP_CHECK_LAUNCH("sep_sisdr_from_dots");
    return 0;
}

extern "C" int sep_sisdr_bwd(const float* est, const float* tgt, const double* dots, const double* tt, const double* xx,
                             const float* gw, float* d_est, int B, int n, int T, int all_pairs, float eps,
                             sep_stream_t stream) {
    SEP_REQUIRE(est && tgt && dots && tt && xx && gw && d_est, "sep_sisdr_bwd: null pointer");
    SEP_REQUIRE(n <= 64 && B <= 65535, "sep_sisdr_bwd: n <= 64 and B <= 65535 supported");
    dim3 grid(ceil_div(T, 1024), n, B);
    hipLaunchKernelGGL(sisdr_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, est, tgt, dots, tt, xx, gw, d_est, n, T, all_pairs, eps);
    SEP_CHECK_LAUNCH("sep_sisdr_bwd");
    return 0;
}

extern "C" int sep_pit_search(const float* val, const int32_t* perms, int P, int n, int B, int maximize, int use_mean,
                              float* best_val, int64_t* best_idx, sep_stream_t stream) {
    SEP_REQUIRE(val && perms && best_val && best_idx && P > 0 && n > 0 && B > 0, "sep_pit_search: bad arguments");
    hipLaunchKernelGGL(pit_search_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, val, perms, P, n, B, maximize, use_mean, best_val, best_idx);
    SEP_CHECK_LAUNCH("sep_pit_search");
    return 0;
}

extern "C" int sep_sinkhorn_fwd(const float* C, double* zwork, float* loss, float* P, int B, int n, float coldness,
                                int iters, sep_stream_t stream) {
    SEP_REQUIRE(C && zwork && loss && P && B > 0 && n > 0 && n <= 32 && iters >= 0 && coldness != 0.f, "sep_sinkhorn_fwd: bad arguments (n <= 32)");
    const size_t smem = (size_t)(n * n + n + 64) * sizeof(double);
    hipLaunchKernelGGL(sinkhorn_fwd_kernel, dim3(B), dim3(64), smem, (hipStream_t)stream, C, zwork, loss, P, n, coldness, iters);
    SEP_CHECK_LAUNCH("sep_sinkhorn_fwd");
    return 0;
}

extern "C" int sep_sinkhorn_bwd(const float* C, const double* zwork, const float* dloss, float* dC, int B, int n,
                                float coldness, int iters, sep_stream_t stream) {
    SEP_REQUIRE(C && zwork && dloss && dC && B > 0 && n > 0 && n <= 32 && iters >= 0, "sep_sinkhorn_bwd: bad arguments (n <= 32)");
    const size_t smem = (size_t)(n * n + n) * sizeof(double);
    hipLaunchKernelGGL(sinkhorn_bwd_kernel, dim3(B), dim3(64), smem, (hipStream_t)stream, C, zwork, dloss, dC, n, coldness, iters);
    SEP_CHECK_LAUNCH("sep_sinkhorn_bwd");
    return 0;
}

extern "C" int sep_rowdiff_sums(const float* x, const float* t, double* sums, int64_t rows, int T, sep_stream_t stream) {
    SEP_REQUIRE(x && t && sums && rows > 0 && rows < (1ll << 31) && T > 0, "sep_rowdiff_sums: bad arguments");
    const bool vec = T % 4 == 0 && (((uintptr_t)x | (uintptr_t)t) & 15) == 0;
    if (vec) hipLaunchKernelGGL(rowdiff_sums_kernel<true>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, t, sums, T);
    else hipLaunchKernelGGL(rowdiff_sums_kernel<false>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, t, sums, T);
    SEP_CHECK_LAUNCH("sep_rowdiff_sums");
    return 0;
}

extern "C" int sep_rowdiff_bwd(const float* x, const float* t, const float* c_abs, const float* c_sq, float* dx, int64_t rows,
                               int T, sep_stream_t stream) {
    SEP_REQUIRE(x && t && dx && (c_abs || c_sq) && rows > 0 && rows < 65536 && T > 0, "sep_rowdiff_bwd: bad arguments");
    const bool vec = T % 4 == 0 && (((uintptr_t)x | (uintptr_t)t | (uintptr_t)dx) & 15) == 0;
    const dim3 grid((unsigned)((T + 1023) / 1024), (unsigned)rows);
    if (vec) hipLaunchKernelGGL(rowdiff_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, t, c_abs, c_sq, dx, T);
    else hipLaunchKernelGGL(rowdiff_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, t, c_abs, c_sq, dx, T);
    SEP_CHECK_LAUNCH("sep_rowdiff_bwd");
    return 0;
}

extern "C" int sep_sqnorm(const float* g, double* sqnorm, int64_t n, sep_stream_t stream) {
    SEP_REQUIRE(g && sqnorm && n > 0, "sep_sqnorm: bad arguments");
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 256) blocks = 256;      // one fp64 atomic per workgroup on ONE address: 2048 of them serialised into ~30 us (profiles/r08a: 33 us for 20 MB)
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, sqnorm, n);
    SEP_CHECK_LAUNCH("sep_sqnorm");
    return 0;
}

extern "C" int sep_adam_step(float* p, float* g, float* m, float* v, const double* sqnorm, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float max_norm, float grad_scale, int step,
                             sep_stream_t stream) {
    SEP_REQUIRE(p && g && m && v && n > 0 && step >= 1, "sep_adam_step: bad arguments");
    SEP_REQUIRE(max_norm <= 0.f || sqnorm, "sep_adam_step: clipping needs sqnorm");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sqnorm, n, lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale, bc1, bc2s);
    SEP_CHECK_LAUNCH("sep_adam_step");
    return 0;
}

// ---- error plumbing ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void sep_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sep_last_error(void) { return g_err; }
extern "C" int sep_version(void) { return SEP_ABI_VERSION; }
