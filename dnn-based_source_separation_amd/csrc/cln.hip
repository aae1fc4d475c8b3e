// Cumulative layer normalisation (causal cLN) of the Conv-TasNet family, forward and backward.
// Replaces CumulativeLayerNorm1d.forward of reference src/modules/norm.py:58-101 and autograd's backward of it:
//   y[b][c][t] = (x[b][c][t] - m_t) * r_t * gamma_c + beta_c,   n_t = C*(t+1),
//   m_t = S1_t / n_t,  v_t = S2_t / n_t - m_t^2,  r_t = 1 / (sqrt(v_t) + eps),  S1_t / S2_t = sums of x / x^2 over all
//   channels and all frames <= t.
// Three launches each way, all HBM-bound streaming passes over (B, C, ldt) with frames contiguous:
//   forward : column sums over the channels (fp32 per frame -> fp64) | per-sample prefix sums in fp64 -> m_t, r_t | apply
//   backward: column sums A_t = sum_c g*gamma, Bq_t = sum_c g*gamma*(x - m_t) | per-sample SUFFIX sums
//             P_t = sum_{t'>=t} Dm_t'/n_t', Q_t = sum_{t'>=t} Dq_t'/n_t' with Dq = -Bq*r^2/(2 sigma), Dm = -r*A - 2 m Dq |
//             dx = g*gamma*r_t + P_t + 2 x Q_t together with the per-(sample, channel) sums for d(gamma), d(beta)
// The prefix sums are taken in fp64 (the reference's fp32 cumsum over thousands of frames followed by E[x^2] - m^2 loses
// digits the fp64 oracle keeps); everything elementwise is fp32.
#include "common.hpp"

namespace {

constexpr int CLN_TCOLS = 64;        // frames per column-sum block (one lane each), 4 waves share the channels

// ws[b][0][t], ws[b][1][t] <- the two column sums of frame t
// alpha (may be NULL): the single slope of a PReLU in front of the norm -- the kernels then normalise u = PReLU(x; alpha) and read / write
// x itself (reference tdcn.py:113-116, 182-186: nonlinear1d then norm1d), one H-tensor round trip less than a stand-alone activation
template <bool BWD>
__global__ __launch_bounds__(256) void cln_colsums_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean, const float* __restrict__ alpha, double* __restrict__ ws, int C, int T, int ldt) {
    __shared__ float red[2][4][CLN_TCOLS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y, t = blockIdx.x * CLN_TCOLS + lane;
    const bool live = t < T;
    const size_t base = (size_t)b * C * ldt + (live ? t : 0);
    const float m = BWD && live ? mean[(size_t)b * ldt + t] : 0.f;
    const bool act = alpha != nullptr;
    const float al = act ? alpha[0] : 1.f;
    float s0 = 0.f, s1 = 0.f;
    for (int c = w; c < C; c += 16) {                   // four rows per trip and wave: independent loads in flight
        float xv[4], gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cc = c + 4 * q;
            xv[q] = cc < C ? x[base + (size_t)cc * ldt] : 0.f;
            if (act) xv[q] = prelu_f(xv[q], al);
            gv[q] = BWD && cc < C ? g[base + (size_t)cc * ldt] * gamma[cc] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (BWD) { s0 += gv[q]; s1 += gv[q] * (xv[q] - m); }
            else { s0 += xv[q]; s1 += xv[q] * xv[q]; }
        }
    }
    red[0][w][lane] = s0;
    red[1][w][lane] = s1;
    __syncthreads();
    if (w == 0 && live) {
        const double a0 = (double)red[0][0][lane] + (double)red[0][1][lane] + (double)red[0][2][lane] + (double)red[0][3][lane];
        const double a1 = (double)red[1][0][lane] + (double)red[1][1][lane] + (double)red[1][2][lane] + (double)red[1][3][lane];
        ws[((size_t)b * 2 + 0) * ldt + t] = a0;
        ws[((size_t)b * 2 + 1) * ldt + t] = a1;
    }
}

// inclusive scan of two doubles over the 1024 threads of a block (thread order = scan order), carry added by the caller
__device__ __forceinline__ void block_scan2(double& a, double& b, double (*sm)[16], double& tot_a, double& tot_b) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ua = __shfl_up(a, off), ub = __shfl_up(b, off);
        if (lane >= off) { a += ua; b += ub; }
    }
    if (lane == 63) { sm[0][w] = a; sm[1][w] = b; }
    __syncthreads();
    double pa = 0.0, pb = 0.0, ta = 0.0, tb = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < w) { pa += sm[0][i]; pb += sm[1][i]; }
        ta += sm[0][i];
        tb += sm[1][i];
    }
    __syncthreads();
    a += pa;
    b += pb;
    tot_a = ta;
    tot_b = tb;
}

// forward: prefix sums -> mean, rstd.   one block per sample
__global__ __launch_bounds__(1024) void cln_scan_fwd_kernel(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ rstd, int C, int T, int ldt, float eps) {
    __shared__ double sm[2][16];
    const int b = blockIdx.x;
    double ca = 0.0, cb = 0.0;
    for (int t0 = 0; t0 < T; t0 += 1024) {
        const int t = t0 + threadIdx.x;
        double a = t < T ? ws[((size_t)b * 2 + 0) * ldt + t] : 0.0, q = t < T ? ws[((size_t)b * 2 + 1) * ldt + t] : 0.0, ta, tb;
        block_scan2(a, q, sm, ta, tb);
        a += ca;
        q += cb;
        ca += ta;
        cb += tb;
        if (t < T) {
            const double n = (double)C * (double)(t + 1);
            const double m = a / n;
            double var = q / n - m * m;
            if (var < 0.0) var = 0.0;
            mean[(size_t)b * ldt + t] = (float)m;
            rstd[(size_t)b * ldt + t] = (float)(1.0 / (sqrt(var) + (double)eps));
        }
    }
}

// backward: ws holds A_t, Bq_t; leaves P_t, Q_t (suffix sums) in their place.   one block per sample, frames visited from the end
__global__ __launch_bounds__(1024) void cln_scan_bwd_kernel(double* __restrict__ ws, const float* __restrict__ mean, const float* __restrict__ rstd, int C, int T, int ldt, float eps) {
    __shared__ double sm[2][16];
    const int b = blockIdx.x;
    double ca = 0.0, cb = 0.0;
    for (int r0 = 0; r0 < T; r0 += 1024) {
        const int t = T - 1 - (r0 + (int)threadIdx.x);                 // reversed: thread order = descending frames
        double dm = 0.0, dq = 0.0;
        if (t >= 0) {
            const double A = ws[((size_t)b * 2 + 0) * ldt + t], Bq = ws[((size_t)b * 2 + 1) * ldt + t];
            const double r = (double)rstd[(size_t)b * ldt + t], m = (double)mean[(size_t)b * ldt + t];
            const double sigma = 1.0 / r - (double)eps;
            const double n = (double)C * (double)(t + 1);
            const double Dq = sigma > 0.0 ? -Bq * r * r / (2.0 * sigma) : 0.0;      // d r / d v = -r^2 / (2 sigma); a constant prefix (sigma = 0) has no slope here
            const double Dm = -r * A - 2.0 * m * Dq;
            dm = Dm / n;
            dq = Dq / n;
        }
        double ta, tb;
        block_scan2(dm, dq, sm, ta, tb);
        dm += ca;
        dq += cb;
        ca += ta;
        cb += tb;
        if (t >= 0) {
            ws[((size_t)b * 2 + 0) * ldt + t] = dm;
            ws[((size_t)b * 2 + 1) * ldt + t] = dq;
        }
    }
}

// one wave per row (b, c): 256 frames per trip
__global__ __launch_bounds__(256) void cln_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ alpha,
                                                            float* __restrict__ y, int C, int T, int ldt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + w, b = blockIdx.y;
    if (c >= C) return;
    const float ga = gamma[c], be = beta[c];
    const bool act = alpha != nullptr;
    const float al = act ? alpha[0] : 1.f;
    const size_t row = ((size_t)b * C + c) * ldt;
    for (int t = 4 * lane; t < ldt; t += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(x + row + t);
        float xs[4] = {xv.x, xv.y, xv.z, xv.w}, o[4];
        // the frames' statistics: rows of ldt floats, so a lane's four frames are ONE 16-byte load (one scalar load per frame made these
        // kernels issue-bound: 175 us per backward apply at the paper-best sizes, profiles/r05o_causal_kernel_stats.md)
        const float4 m4 = *reinterpret_cast<const float4*>(mean + (size_t)b * ldt + t), r4 = *reinterpret_cast<const float4*>(rstd + (size_t)b * ldt + t);
        const float ms[4] = {m4.x, m4.y, m4.z, m4.w}, rs[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool live = t + e < T;
            const float m = live ? ms[e] : 0.f, r = live ? rs[e] : 0.f;
            const float u = act ? prelu_f(xs[e], al) : xs[e];
            o[e] = live ? (u - m) * r * ga + be : 0.f;
        }
        *reinterpret_cast<float4*>(y + row + t) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void cln_apply_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const double* __restrict__ ws, const float* __restrict__ gamma, const float* __restrict__ alpha, float* __restrict__ dx,
                                                            float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, float* __restrict__ dalpha_part, int C, int T, int ldt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + w, b = blockIdx.y;
    if (c >= C) return;
    const float ga = gamma[c];
    const bool act = alpha != nullptr;
    const float al = act ? alpha[0] : 1.f;
    const size_t row = ((size_t)b * C + c) * ldt;
    double sg = 0.0, sb = 0.0, sa = 0.0;
    for (int t = 4 * lane; t < ldt; t += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(x + row + t), gv = *reinterpret_cast<const float4*>(g + row + t);
        float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, o[4];
        float pg = 0.f, pb = 0.f, pa = 0.f;
        const float4 m4 = *reinterpret_cast<const float4*>(mean + (size_t)b * ldt + t), r4 = *reinterpret_cast<const float4*>(rstd + (size_t)b * ldt + t);
        const double2 p01 = *reinterpret_cast<const double2*>(ws + ((size_t)b * 2 + 0) * ldt + t), p23 = *reinterpret_cast<const double2*>(ws + ((size_t)b * 2 + 0) * ldt + t + 2);
        const double2 q01 = *reinterpret_cast<const double2*>(ws + ((size_t)b * 2 + 1) * ldt + t), q23 = *reinterpret_cast<const double2*>(ws + ((size_t)b * 2 + 1) * ldt + t + 2);
        const float ms[4] = {m4.x, m4.y, m4.z, m4.w}, rs[4] = {r4.x, r4.y, r4.z, r4.w};
        const float Ps[4] = {(float)p01.x, (float)p01.y, (float)p23.x, (float)p23.y}, Qs[4] = {(float)q01.x, (float)q01.y, (float)q23.x, (float)q23.y};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool live = t + e < T;
            const float m = live ? ms[e] : 0.f, r = live ? rs[e] : 0.f;
            const float P = live ? Ps[e] : 0.f, Q = live ? Qs[e] : 0.f;
            const float gl = live ? gs[e] : 0.f;
            const float u = act ? prelu_f(xs[e], al) : xs[e];
            const float du = live ? gl * ga * r + P + 2.f * u * Q : 0.f;
            o[e] = act ? du * prelu_grad(xs[e], al) : du;
            if (act && xs[e] <= 0.f) pa = fmaf(du, xs[e], pa);
            pg += gl * (u - m) * r;
            pb += gl;
        }
        sg += (double)pg;
        sb += (double)pb;
        sa += (double)pa;
        *reinterpret_cast<float4*>(dx + row + t) = make_float4(o[0], o[1], o[2], o[3]);
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if (act) sa = wave_sum(sa);
    if (lane == 0) {
        dgamma_part[(size_t)b * C + c] = (float)sg;
        dbeta_part[(size_t)b * C + c] = (float)sb;
        if (act) dalpha_part[(size_t)b * C + c] = (float)sa;
    }
}

}  // namespace

extern "C" int sep_cln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, double* ws,
                           int B, int C, int T, int ldt, float eps, const float* alpha, sep_stream_t stream_) {
    SEP_REQUIRE(x && gamma && beta && y && mean && rstd && ws && B > 0 && B <= 65535 && C > 0 && T > 0 && ldt >= T && ldt % 4 == 0, "sep_cln_fwd: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL((cln_colsums_kernel<false>), dim3(ceil_div(T, CLN_TCOLS), B), dim3(256), 0, stream, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, alpha, ws, C, T, ldt);
    hipLaunchKernelGGL(cln_scan_fwd_kernel, dim3(B), dim3(1024), 0, stream, (const double*)ws, mean, rstd, C, T, ldt, eps);
    hipLaunchKernelGGL(cln_apply_fwd_kernel, dim3(ceil_div(C, 4), B), dim3(256), 0, stream, x, (const float*)mean, (const float*)rstd, gamma, beta, alpha, y, C, T, ldt);
    SEP_CHECK_LAUNCH("sep_cln_fwd");
    return 0;
}

extern "C" int sep_cln_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                           float* dgamma_part, float* dbeta_part, double* ws, int B, int C, int T, int ldt, float eps, const float* alpha,
                           float* dalpha_part, sep_stream_t stream_) {
    SEP_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma_part && dbeta_part && ws && B > 0 && B <= 65535 && C > 0 && T > 0 && ldt >= T && ldt % 4 == 0,
                "sep_cln_bwd: bad arguments");
    SEP_REQUIRE((alpha == nullptr) == (dalpha_part == nullptr), "sep_cln_bwd: alpha and dalpha_part come together");
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL((cln_colsums_kernel<true>), dim3(ceil_div(T, CLN_TCOLS), B), dim3(256), 0, stream, x, dy, gamma, mean, alpha, ws, C, T, ldt);
    hipLaunchKernelGGL(cln_scan_bwd_kernel, dim3(B), dim3(1024), 0, stream, ws, mean, rstd, C, T, ldt, eps);
    hipLaunchKernelGGL(cln_apply_bwd_kernel, dim3(ceil_div(C, 4), B), dim3(256), 0, stream, dy, x, mean, rstd, (const double*)ws, gamma, alpha, dx, dgamma_part, dbeta_part, dalpha_part, C, T, ldt);
    SEP_CHECK_LAUNCH("sep_cln_bwd");
    return 0;
}
