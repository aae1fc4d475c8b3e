// HBM-bound (streaming) kernels of the Conv-TasNet path for gfx950: learned encoder / decoder with
// overlap-add, dilated depthwise conv with the gLN + PReLU of its input fused on load (LDS-staged
// receptive field), their backward forms, the second stage of the gLN backward reductions, and a
// stand-alone gLN.  All tensors (batch, channel, frame) fp32 with padded row stride ldt; frames [T, ldt)
// of every written tensor are zero.
//
// Reference arithmetic replaced (under /root/reference/src): models/filterbank.py:205-251 (Encoder/Decoder),
// models/conv_tasnet.py:145-169 (pad, mask*w, crop), models/tdcn.py:113-132,177-186 (PReLU -> gLN -> zero pad ->
// depthwise conv -> PReLU), modules/norm.py:11-35 (gLN = GroupNorm(1, C)).
#include <stdlib.h>
#include "common.hpp"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// =====================================================================================
// Encoder: w[b][n][f] = sum_{c,k} E[n][c][k] * xpad[b][c][S f + k]   (+ReLU), + gLN statistics of w
// One block = 128 frames x all N basis rows.  Thread = (frame, n parity); the basis row is wave-uniform
// (scalar loads), the frame's samples sit in registers (LC = Cin*L <= 16) or LDS (generic).
// =====================================================================================
constexpr int ENC_FT = 128;

template <int LC_REG>
__global__ __launch_bounds__(256) void encoder_fwd_kernel(const float* __restrict__ x, const float* __restrict__ E,
                                                          float* __restrict__ w, double* __restrict__ stats, int B,
                                                          int Cin, int Tin, int N, int L, int S, int F, int ldt,
                                                          int pad_left, int relu) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [Cin][span]
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * ENC_FT;
    const int span = ENC_FT * S + L - S;
    for (int i = tid; i < Cin * span; i += 256) {
        const int c = i / span, o = i % span;
        const int tau = S * f0 + o - pad_left;
        xs[i] = (tau >= 0 && tau < Tin) ? x[((size_t)b * Cin + c) * Tin + tau] : 0.f;
    }
    __syncthreads();
    const int fl = tid & (ENC_FT - 1);
    const int f = f0 + fl;
    const int nh = __builtin_amdgcn_readfirstlane(tid >> 7);   // wave-uniform
    const bool fvalid = f < F;
    const int LC = Cin * L;
    float xr[LC_REG > 0 ? LC_REG : 1];
    if (LC_REG > 0) {
#pragma unroll
        for (int q = 0; q < LC_REG; ++q) {
            const int c = q / L, k = q % L;
            xr[q] = xs[c * span + S * fl + k];
        }
    }
    float s = 0.f, ss = 0.f;
    for (int n = nh; n < N; n += 2) {
        const float* En = E + (size_t)n * LC;
        float acc = 0.f;
        if (LC_REG > 0) {
#pragma unroll
            for (int q = 0; q < LC_REG; ++q) acc = fmaf(En[q], xr[q], acc);
        } else {
            for (int c = 0; c < Cin; ++c)
                for (int k = 0; k < L; ++k) acc = fmaf(En[c * L + k], xs[c * span + S * fl + k], acc);
        }
        if (relu) acc = fmaxf(acc, 0.f);
        if (!fvalid) acc = 0.f;
        s += acc; ss += acc * acc;
        if (f < ldt) w[((size_t)b * N + n) * ldt + f] = acc;
    }
    const double ds = block_sum_256<double>((double)s, red);
    const double dss = block_sum_256<double>((double)ss, red);
    if (tid == 0) { double* st = stats + ((size_t)b * SEP_STATS_SLOTS + (blockIdx.x & (SEP_STATS_SLOTS - 1))) * 2; atomicAdd(st, ds); atomicAdd(st + 1, dss); }
}

// Mono 16-tap / stride-8 encoder (the paper-best front end): a lane owns FOUR consecutive frames (40 input samples in registers),
// so every store of w is a float4 -- 1 KiB per wave instruction instead of 256 B (the dword-per-lane form is store-issue bound:
// 81 us for 131 MB).  One workgroup = 256 frames of one sample; its four waves take the basis rows n = wave (mod 4).
__global__ __launch_bounds__(256) void encoder_fwd_l16s8_kernel(const float* __restrict__ x, const float* __restrict__ E, float* __restrict__ w,
                                                                double* __restrict__ stats, int Tin, int N, int F, int ldt, int pad_left, int relu) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int f = blockIdx.x * 256 + 4 * lane;            // first of this lane's four frames
    float xr[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) {
        const int tau = 8 * f + i - pad_left;
        xr[i] = (tau >= 0 && tau < Tin) ? x[(size_t)b * Tin + tau] : 0.f;
    }
    float s = 0.f, ss = 0.f;
    if (f < ldt) {
        for (int n = wv; n < N; n += 4) {
            const float* En = E + (size_t)n * 16;         // wave-uniform: scalar loads
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fmaf(En[k], xr[8 * j + k], acc);
                if (relu) acc = fmaxf(acc, 0.f);
                if (f + j >= F) acc = 0.f;
                s += acc; ss = fmaf(acc, acc, ss);
                o[j] = acc;
            }
            st4(w + ((size_t)b * N + n) * ldt + f, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
    const double ds = block_sum_256<double>((double)s, red);
    const double dss = block_sum_256<double>((double)ss, red);
    if (threadIdx.x == 0) { double* st = stats + ((size_t)b * SEP_STATS_SLOTS + (blockIdx.x & (SEP_STATS_SLOTS - 1))) * 2; atomicAdd(st, ds); atomicAdd(st + 1, dss); }
}

// frames[bp][c*L+k][f] = xpad[bp][c][S f + k] (f < F), 0 for F <= f < ldt
__global__ __launch_bounds__(256) void unfold_kernel(const float* __restrict__ x, float* __restrict__ frames, int C,
                                                     int Tin, int L, int S, int F, int ldt, int pad_left) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int row = blockIdx.y;           // c*L + k
    const int bp = blockIdx.z;
    if (f >= ldt) return;
    const int c = row / L, k = row % L;
    const int tau = S * f + k - pad_left;
    float v = 0.f;
    if (f < F && tau >= 0 && tau < Tin) v = x[((size_t)bp * C + c) * Tin + tau];
    frames[((size_t)bp * C * L + row) * ldt + f] = v;
}

// =====================================================================================
// Depthwise dilated conv, forward.  One wave = one (b, c, 1024-frame tile): the tile plus a halo of
// round_up(d,4) frames each side is loaded with float4, normalised (gLN o PReLU) and zero-masked on the
// way into wave-private LDS; outputs are computed lane-consecutive (conflict-free ds_read_b32).
// =====================================================================================
constexpr int DW_TT = 1024;

__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ a, const double* __restrict__ stats1,
                                                         const float* __restrict__ gamma1, const float* __restrict__ beta1,
                                                         const float* __restrict__ alpha1, const float* __restrict__ wd,
                                                         const float* __restrict__ bd, const float* __restrict__ alpha2,
                                                         float* __restrict__ z, double* __restrict__ stats2, int B, int C,
                                                         int T, int ldt, int d, int dpad, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double wred[4][2];
    __shared__ int wb[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntile = (ldt + DW_TT - 1) / DW_TT;
    const long total = (long)B * C * ntile;
    const long g = (long)blockIdx.x * 4 + wv;
    const bool active = g < total;
    const int wlen = DW_TT + 2 * dpad;
    float* vs = lds + (size_t)wv * wlen;

    int b = 0, c = 0, t0 = 0;
    float s = 0.f, ss = 0.f;
    if (active) {
        const int tile = (int)(g % ntile);
        c = (int)((g / ntile) % C);
        b = (int)(g / ((long)ntile * C));
        t0 = tile * DW_TT;
        float mu, rstd;
        gln_mu_rstd(stats1 + (size_t)b * SEP_STATS_SLOTS * 2, (double)C * T, eps, mu, rstd);
        const float a1 = alpha1[0];
        const float sc = gamma1[c] * rstd, sh = beta1[c] - mu * sc;
        const float* arow = a + ((size_t)b * C + c) * ldt;
        for (int q = lane; q < wlen / 4; q += 64) {
            const int tp = t0 - dpad + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tp >= 0 && tp < ldt) {
                const float4 r = ld4(arow + tp);
                v.x = (tp + 0 < T) ? prelu_f(r.x, a1) * sc + sh : 0.f;
                v.y = (tp + 1 < T) ? prelu_f(r.y, a1) * sc + sh : 0.f;
                v.z = (tp + 2 < T) ? prelu_f(r.z, a1) * sc + sh : 0.f;
                v.w = (tp + 3 < T) ? prelu_f(r.w, a1) * sc + sh : 0.f;
            }
            st4(vs + 4 * q, v);
        }
    }
    __syncthreads();
    if (active) {
        const float w0 = wd[c * 3 + 0], w1 = wd[c * 3 + 1], w2 = wd[c * 3 + 2], bb = bd[c];
        const float a2 = alpha2[0];
        float* zrow = z + ((size_t)b * C + c) * ldt;
#pragma unroll 4
        for (int i = 0; i < DW_TT / 64; ++i) {
            const int o = lane + 64 * i;
            const int t = t0 + o;
            if (t >= ldt) break;
            float zz = 0.f;
            if (t < T) {
                zz = bb + w0 * vs[dpad + o - d] + w1 * vs[dpad + o] + w2 * vs[dpad + o + d];
                const float u = prelu_f(zz, a2);
                s += u; ss += u * u;
            }
            zrow[t] = zz;
        }
    }
    const double dsum = wave_sum((double)s), dss = wave_sum((double)ss);
    if (lane == 0) { wred[wv][0] = dsum; wred[wv][1] = dss; wb[wv] = active ? b : -1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // merge waves that belong to the same sample -> at most one atomic pair per sample per block
        for (int i = 0; i < 4; ++i) {
            if (wb[i] < 0) continue;
            double x0 = wred[i][0], x1 = wred[i][1];
            for (int k = i + 1; k < 4; ++k)
                if (wb[k] == wb[i]) { x0 += wred[k][0]; x1 += wred[k][1]; wb[k] = -1; }
            double* st = stats2 + ((size_t)wb[i] * SEP_STATS_SLOTS + (blockIdx.x & (SEP_STATS_SLOTS - 1))) * 2;
            atomicAdd(st, x0);
            atomicAdd(st + 1, x1);
        }
    }
}

// =====================================================================================
// Depthwise dilated conv forward, direct form (dilation 1, 2 or a multiple of 4 -- every power of two, i.e. every
// layer of the TCN): one workgroup = one (b, c) row; a thread produces float4s of output from three float4 loads
// (t-d, t, t+d; for d < 4 the two neighbouring float4s and a register shuffle).  The neighbours are another thread's
// centre, so two of the three loads hit L1/L2; nothing goes through LDS and there is no barrier before the stores.
// (Issuing all of a thread's loads before the first use, as dwconv_bwd_row_kernel does, was measured 6 % SLOWER here.)
// DM: 0 -> d % 4 == 0, 1 -> d == 1, 2 -> d == 2.
// =====================================================================================
template <int DM, int ROWS>
__global__ __launch_bounds__(256) void dwconv_fwd_direct_kernel(const float* __restrict__ a, const double* __restrict__ stats1,
                                                                const float* __restrict__ gamma1, const float* __restrict__ beta1,
                                                                const float* __restrict__ alpha1, const float* __restrict__ wd,
                                                                const float* __restrict__ bd, const float* __restrict__ alpha2,
                                                                float* __restrict__ z, double* __restrict__ stats2, int C, int T,
                                                                int ldt, int d, float eps) {
    __shared__ double red[4];
    const int row0 = blockIdx.x * ROWS;       // b * C + c of the first of ROWS consecutive channels of one sample (C % ROWS == 0)
    const int b = row0 / C;
    float mu, rstd;
    gln_mu_rstd(stats1 + (size_t)b * SEP_STATS_SLOTS * 2, (double)C * T, eps, mu, rstd);
    const float a1 = alpha1[0], a2 = alpha2[0];
    float s = 0.f, ss = 0.f;
#pragma unroll 1
    for (int rr = 0; rr < ROWS; ++rr) {
    const int row = row0 + rr, c = row % C;
    const float sc = gamma1[c] * rstd, sh = beta1[c] - mu * sc;
    const float w0 = wd[c * 3 + 0], w1 = wd[c * 3 + 1], w2 = wd[c * 3 + 2], bb = bd[c];
    const float* arow = a + (size_t)row * ldt;
    float* zrow = z + (size_t)row * ldt;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto norm = [&](float x, int t) { return (t >= 0 && t < T) ? fmaf(prelu_f(x, a1), sc, sh) : 0.f; };
    for (int q = threadIdx.x; q < ldt / 4; q += 256) {
        const int t = 4 * q;
        const float4 cc = ld4(arow + t);
        float lft[4], rgt[4];
        if (DM == 0) {
            const float4 l4 = t - d >= 0 ? ld4(arow + t - d) : zero4;
            const float4 r4 = t + d < ldt ? ld4(arow + t + d) : zero4;
            lft[0] = l4.x; lft[1] = l4.y; lft[2] = l4.z; lft[3] = l4.w;
            rgt[0] = r4.x; rgt[1] = r4.y; rgt[2] = r4.z; rgt[3] = r4.w;
        } else {
            const float4 l4 = t >= 4 ? ld4(arow + t - 4) : zero4;
            const float4 r4 = t + 4 < ldt ? ld4(arow + t + 4) : zero4;
            if (DM == 1) {
                lft[0] = l4.w; lft[1] = cc.x; lft[2] = cc.y; lft[3] = cc.z;
                rgt[0] = cc.y; rgt[1] = cc.z; rgt[2] = cc.w; rgt[3] = r4.x;
            } else {
                lft[0] = l4.z; lft[1] = l4.w; lft[2] = cc.x; lft[3] = cc.y;
                rgt[0] = cc.z; rgt[1] = cc.w; rgt[2] = r4.x; rgt[3] = r4.y;
            }
        }
        const float ce[4] = {cc.x, cc.y, cc.z, cc.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int te = t + e;
            float zz = 0.f;
            if (te < T) {
                zz = bb + w0 * norm(lft[e], te - d) + w1 * norm(ce[e], te) + w2 * norm(rgt[e], te + d);
                const float u = prelu_f(zz, a2);
                s += u; ss = fmaf(u, u, ss);
            }
            o[e] = zz;
        }
        st4(zrow + t, make_float4(o[0], o[1], o[2], o[3]));
    }
    }
    const double ds = block_sum_256<double>((double)s, red);
    const double dss = block_sum_256<double>((double)ss, red);
    if (threadIdx.x == 0) {
        double* st = stats2 + ((size_t)b * SEP_STATS_SLOTS + (blockIdx.x & (SEP_STATS_SLOTS - 1))) * 2;
        atomicAdd(st, ds);
        atomicAdd(st + 1, dss);
    }
}

// =====================================================================================
// Backward of [gLN2 o PReLU2 o depthwise]: dv2 -> dv1 plus the per-row partial sums every downstream
// reduction needs (gLN1 backward, depthwise weight/bias gradient, PReLU2 slope gradient).
// rowpart[b][c][tile][8] = {sum dv1, sum dv1*u1, sum dz, sum dz*v1(t-d), sum dz*v1(t), sum dz*v1(t+d), dalpha2, 0}
// =====================================================================================
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(
    const float* __restrict__ dv2, const float* __restrict__ z, const float* __restrict__ a,
    const double* __restrict__ stats1, const float* __restrict__ gamma1, const float* __restrict__ beta1,
    const float* __restrict__ alpha1, const double* __restrict__ stats2, const float* __restrict__ gamma2,
    const float* __restrict__ alpha2, const float* __restrict__ bsum2, const float* __restrict__ wd,
    float* __restrict__ dv1, float* __restrict__ rowpart, double* __restrict__ bacc1, int* __restrict__ arrive1, float* __restrict__ bsum1,
    int B, int C, int T, int ldt, int d, int dpad, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntile = (ldt + DW_TT - 1) / DW_TT;
    const long total = (long)B * C * ntile;
    const long g = (long)blockIdx.x * 4 + wv;
    const bool active = g < total;
    const int wlen = DW_TT + 2 * dpad;
    float* dzs = lds + (size_t)wv * 2 * wlen;
    float* us = dzs + wlen;

    int b = 0, c = 0, t0 = 0, tile = 0;
    float sc1 = 0.f, sh1 = 0.f;
    float q_dal = 0.f;
    if (active) {
        tile = (int)(g % ntile);
        c = (int)((g / ntile) % C);
        b = (int)(g / ((long)ntile * C));
        t0 = tile * DW_TT;
        float mu1, r1, mu2, r2;
        gln_mu_rstd(stats1 + (size_t)b * SEP_STATS_SLOTS * 2, (double)C * T, eps, mu1, r1);
        gln_mu_rstd(stats2 + (size_t)b * SEP_STATS_SLOTS * 2, (double)C * T, eps, mu2, r2);
        const float a1 = alpha1[0], a2 = alpha2[0];
        sc1 = gamma1[c] * r1; sh1 = beta1[c] - mu1 * sc1;
        const float g2 = gamma2[c];
        const float mg = bsum2[2 * b], mgx = bsum2[2 * b + 1];
        const size_t rowoff = ((size_t)b * C + c) * ldt;
        for (int q = lane; q < wlen / 4; q += 64) {
            const int tp = t0 - dpad + 4 * q;
            float dzv[4] = {0.f, 0.f, 0.f, 0.f}, uv[4] = {0.f, 0.f, 0.f, 0.f};
            if (tp >= 0 && tp < ldt) {
                const float4 gv = ld4(dv2 + rowoff + tp);
                const float4 zv = ld4(z + rowoff + tp);
                const float4 av = ld4(a + rowoff + tp);
                const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
                const float z4[4] = {zv.x, zv.y, zv.z, zv.w};
                const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = tp + e;
                    if (t < T) {
                        const float u2 = prelu_f(z4[e], a2);
                        const float xh = (u2 - mu2) * r2;
                        const float du2 = r2 * (g2 * g4[e] - mg - xh * mgx);
                        dzv[e] = du2 * prelu_grad(z4[e], a2);
                        uv[e] = prelu_f(a4[e], a1);
                        if (z4[e] <= 0.f && t >= t0 && t < t0 + DW_TT) q_dal += du2 * z4[e];
                    }
                }
            }
            st4(dzs + 4 * q, make_float4(dzv[0], dzv[1], dzv[2], dzv[3]));
            st4(us + 4 * q, make_float4(uv[0], uv[1], uv[2], uv[3]));
        }
    }
    __syncthreads();
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f;
    if (active) {
        const float w0 = wd[c * 3 + 0], w1 = wd[c * 3 + 1], w2 = wd[c * 3 + 2];
        float* orow = dv1 + ((size_t)b * C + c) * ldt;
#pragma unroll 4
        for (int i = 0; i < DW_TT / 64; ++i) {
            const int o = lane + 64 * i;
            const int t = t0 + o;
            if (t >= ldt) break;
            float dv = 0.f;
            if (t < T) {
                const int p = dpad + o;
                // dv1[t] = sum_k w[k] * dz[t - (k-1) d]
                dv = w0 * dzs[p + d] + w1 * dzs[p] + w2 * dzs[p - d];
                const float dzt = dzs[p];
                // v1 = gLN1(u1) inside [0,T), literal zero outside (padding is applied after the norm)
                const float vm = (t - d >= 0) ? us[p - d] * sc1 + sh1 : 0.f;
                const float v0 = us[p] * sc1 + sh1;
                const float vp = (t + d < T) ? us[p + d] * sc1 + sh1 : 0.f;
                q0 += dv; q1 += dv * us[p];
                q2 += dzt; q3 += dzt * vm; q4 += dzt * v0; q5 += dzt * vp;
            }
            orow[t] = dv;
        }
    }
    q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2); q3 = wave_sum(q3);
    q4 = wave_sum(q4); q5 = wave_sum(q5); q_dal = wave_sum(q_dal);
    if (active && lane == 0) {
        float* rp = rowpart + (((size_t)b * C + c) * ntile + tile) * 8;
        rp[0] = q0; rp[1] = q1; rp[2] = q2; rp[3] = q3; rp[4] = q4; rp[5] = q5; rp[6] = q_dal; rp[7] = 0.f;
        if (bacc1) {        // gLN1's gamma-weighted totals; the sample's last (channel, tile) publishes the two means (gln_bwd_publish)
            const float g1 = gamma1[c];
            const int slot = (int)(g & (SEP_STATS_SLOTS - 1));
            // slots are taken by the global unit number g; a sample's units are C * ntile consecutive numbers starting at b * C * ntile
            const long first = (long)b * C * ntile, nun = (long)C * ntile;
            int in_slot = 0, used = 0;
            for (int q = 0; q < SEP_STATS_SLOTS; ++q) {
                const long f0 = first + ((q - first) & (SEP_STATS_SLOTS - 1));      // first unit of the sample landing in slot q
                const int cnt_q = f0 < first + nun ? (int)((first + nun - f0 + SEP_STATS_SLOTS - 1) / SEP_STATS_SLOTS) : 0;
                used += cnt_q > 0;
                if (q == slot) in_slot = cnt_q;
            }
            if (arrive1)
                gln_bwd_publish(bacc1 + (size_t)b * SEP_STATS_SLOTS * 2, stats1 + (size_t)b * SEP_STATS_SLOTS * 2, arrive1 + (size_t)b * SEP_ARRIVE_INTS,
                                bsum1 + 2 * b, slot, in_slot, used, (double)(g1 * q0), (double)(g1 * q1), (double)C * T, eps);
            else {      // sums only: the consumer forms the means (sep_gemm_desc.pro_bacc)
                double* ba = bacc1 + ((size_t)b * SEP_STATS_SLOTS + slot) * 2;
                atomicAdd(ba, (double)(g1 * q0)); atomicAdd(ba + 1, (double)(g1 * q1));
            }
        }
    }
}

// =====================================================================================
// Same backward, one workgroup per (b, c) ROW: dz of the whole row lives in LDS (ldt floats), so there is no halo to re-load
// or re-compute (the 1024-frame tiles of the kernel above re-do 25 % of the row at dilation 128) and every global access is
// a float4.  u1 never goes to LDS: the three correlation sums are re-indexed onto the thread's own frame,
//   sum_t dz[t] v1[t-d] = sum_s v1[s] dz[s+d],   sum_t dz[t] v1[t+d] = sum_s v1[s] dz[s-d]      (dz = 0 outside [0, T)),
// so a thread keeps the PReLU1 outputs of its NIT float4 in registers across the barrier: 16 KiB of LDS per workgroup at
// ldt = 4096 instead of 32 (8 workgroups per CU instead of 4) and half the LDS traffic.  DM: see dw_row_neighbours (d % 4 == 0: ds_read_b128
// neighbours); NIT >= ldt / 1024.  Writes the row totals into tile 0 of rowpart and zeros into the other tiles (the
// finalize kernel sums over tiles).
// =====================================================================================
// the neighbours at distance d of the four frames t .. t + 3 of an LDS row: DM 0: d % 4 == 0 (two aligned float4 reads), 1 / 2: d = 1 / 2 (the
// two neighbouring float4s and register selects, as dwconv_fwd_direct_kernel does with global loads), 3: any d (eight scalar reads)
#ifndef DWB_WAVES
#define DWB_WAVES 6      // waves per SIMD the depthwise backward row kernel is compiled for (79 registers, no spills; 8 would spill 34)
#endif
template <int DM>
__device__ __forceinline__ void dw_row_neighbours(const float* row, const float4 c, const int t, const int d, const int ldt, float (&lft)[4], float (&rgt)[4]) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DM == 0) {
        const float4 m1 = t - d >= 0 ? ld4(row + t - d) : zero4, p1 = t + d < ldt ? ld4(row + t + d) : zero4;
        lft[0] = m1.x; lft[1] = m1.y; lft[2] = m1.z; lft[3] = m1.w;
        rgt[0] = p1.x; rgt[1] = p1.y; rgt[2] = p1.z; rgt[3] = p1.w;
    } else if (DM == 1 || DM == 2) {
        const float4 l4 = t >= 4 ? ld4(row + t - 4) : zero4, r4 = t + 4 < ldt ? ld4(row + t + 4) : zero4;
        if (DM == 1) {
            lft[0] = l4.w; lft[1] = c.x; lft[2] = c.y; lft[3] = c.z;
            rgt[0] = c.y; rgt[1] = c.z; rgt[2] = c.w; rgt[3] = r4.x;
        } else {
            lft[0] = l4.z; lft[1] = l4.w; lft[2] = c.x; lft[3] = c.y;
            rgt[0] = c.z; rgt[1] = c.w; rgt[2] = r4.x; rgt[3] = r4.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tm = t + e - d, tq = t + e + d;
            lft[e] = tm >= 0 ? row[tm] : 0.f;
            rgt[e] = tq < ldt ? row[tq] : 0.f;
        }
    }
}

template <int DM, int NIT, bool RECOMP>
__global__ __launch_bounds__(256, NIT <= 4 ? DWB_WAVES : 3) void dwconv_bwd_row_kernel(
    const float* __restrict__ dv2, const float* __restrict__ z, const float* __restrict__ bd, const float* __restrict__ a,
    const double* __restrict__ stats1, const float* __restrict__ gamma1, const float* __restrict__ beta1,
    const float* __restrict__ alpha1, const double* __restrict__ stats2, const float* __restrict__ gamma2,
    const float* __restrict__ alpha2, const float* __restrict__ bsum2, const float* __restrict__ wd,
    float* __restrict__ dv1, float* __restrict__ rowpart, double* __restrict__ bacc1, int* __restrict__ arrive1, float* __restrict__ bsum1,
    int C, int T, int ldt, int d, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float part[4][8];
    float* dzs = lds;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const int b = row / C, c = row % C;
    float mu1, r1, mu2, r2;
    gln_mu_rstd(stats1 + (size_t)b * SEP_STATS_SLOTS * 2, (double)C * T, eps, mu1, r1);
    gln_mu_rstd(stats2 + (size_t)b * SEP_STATS_SLOTS * 2, (double)C * T, eps, mu2, r2);
    const float a1 = alpha1[0], a2 = alpha2[0];
    const float sc1 = gamma1[c] * r1, sh1 = beta1[c] - mu1 * sc1;
    const float g2 = gamma2[c];
    const float mg = bsum2[2 * b], mgx = bsum2[2 * b + 1];
    const size_t rowoff = (size_t)row * ldt;
    const int nq4 = ldt / 4;
    float q_dal = 0.f;
    float4 gv[NIT], zv[NIT], av[NIT];
    float uv[NIT][4];
    const float w0 = wd[c * 3 + 0], w1 = wd[c * 3 + 1], w2 = wd[c * 3 + 2];
    // all global reads of the row first: 3 (RECOMP: 2) * NIT float4 in flight per thread
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int q = threadIdx.x + 256 * k;
        if (q < nq4) {
            gv[k] = ld4(dv2 + rowoff + 4 * q);
            if (!RECOMP) zv[k] = ld4(z + rowoff + 4 * q);
            av[k] = ld4(a + rowoff + 4 * q);
        }
    }
    if (RECOMP) {
        // z is not read: the row of v1 = gLN1(PReLU1(a)) goes to a second LDS row and z = bd + the three taps is formed again exactly as
        // dwconv_fwd_direct_kernel formed it (same expression, zero outside [0, T)) -- a quarter of the kernel's HBM bytes for 8 B of LDS
        // traffic per frame
        // v1s and dzs are never live together (v1 is dead once every thread has formed its z): ONE LDS row serves both, behind one more
        // barrier -- 16 KiB per workgroup at ldt = 4096 instead of 32: six workgroups per compute unit (the register limit) instead of four
        float* v1s = lds;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int q = threadIdx.x + 256 * k;
            if (q < nq4) {
                const int tp = 4 * q;
                const float a4[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uv[k][e] = tp + e < T ? prelu_f(a4[e], a1) : 0.f;
                    v4[e] = tp + e < T ? fmaf(uv[k][e], sc1, sh1) : 0.f;
                }
                st4(v1s + tp, make_float4(v4[0], v4[1], v4[2], v4[3]));
            }
        }
        __syncthreads();
        const float bb = bd[c];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int q = threadIdx.x + 256 * k;
            if (q < nq4) {
                const int t = 4 * q;
                float lft[4], rgt[4], o[4];
                float c4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) c4[e] = t + e < T ? fmaf(uv[k][e], sc1, sh1) : 0.f;      // this thread's own v1 (what it wrote)
                dw_row_neighbours<DM>(v1s, make_float4(c4[0], c4[1], c4[2], c4[3]), t, d, ldt, lft, rgt);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ce = t + e < T ? fmaf(uv[k][e], sc1, sh1) : 0.f;
                    o[e] = t + e < T ? bb + w0 * lft[e] + w1 * ce + w2 * rgt[e] : 0.f;
                }
                zv[k] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();                                  // every neighbour read of v1 is done: the row becomes dz
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int q = threadIdx.x + 256 * k;
        if (q < nq4) {
            const int tp = 4 * q;
            const float g4[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w};
            const float z4[4] = {zv[k].x, zv[k].y, zv[k].z, zv[k].w};
            const float a4[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
            float dzv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dzv[e] = 0.f;
                if (!RECOMP) uv[k][e] = 0.f;
                if (tp + e < T) {
                    const float u2 = prelu_f(z4[e], a2);
                    const float xh = (u2 - mu2) * r2;
                    const float du2 = r2 * (g2 * g4[e] - mg - xh * mgx);
                    dzv[e] = du2 * prelu_grad(z4[e], a2);
                    if (!RECOMP) uv[k][e] = prelu_f(a4[e], a1);
                    if (z4[e] <= 0.f) q_dal = fmaf(du2, z4[e], q_dal);
                }
            }
            st4(dzs + tp, make_float4(dzv[0], dzv[1], dzv[2], dzv[3]));
        }
    }
    __syncthreads();
    float* orow = dv1 + rowoff;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f, q5 = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int q = threadIdx.x + 256 * k;
        if (q < nq4) {
            const int t = 4 * q;
            const float4 dzc4 = ld4(dzs + t);
            float dzm[4], dzp[4];
            dw_row_neighbours<DM>(dzs, dzc4, t, d, ldt, dzm, dzp);
            const float dzc[4] = {dzc4.x, dzc4.y, dzc4.z, dzc4.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int te = t + e;
                float dv = 0.f;
                if (te < T) {
                    // dv1[t] = sum_k w[k] * dz[t - (k-1) d];  dz is zero outside [0, T) by construction
                    dv = w0 * dzp[e] + w1 * dzc[e] + w2 * dzm[e];
                    // v1 = gLN1(u1) inside [0,T) (the zero padding is applied after the norm and is covered by dz = 0 there)
                    const float v0 = fmaf(uv[k][e], sc1, sh1);
                    q0 += dv; q1 = fmaf(dv, uv[k][e], q1);
                    q2 += dzc[e]; q3 = fmaf(dzp[e], v0, q3); q4 = fmaf(dzc[e], v0, q4); q5 = fmaf(dzm[e], v0, q5);
                }
                o[e] = dv;
            }
            st4(orow + t, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
    q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2); q3 = wave_sum(q3);
    q4 = wave_sum(q4); q5 = wave_sum(q5); q_dal = wave_sum(q_dal);
    if (lane == 0) {
        part[wv][0] = q0; part[wv][1] = q1; part[wv][2] = q2; part[wv][3] = q3;
        part[wv][4] = q4; part[wv][5] = q5; part[wv][6] = q_dal; part[wv][7] = 0.f;
    }
    __syncthreads();
    const int ntile = (ldt + DW_TT - 1) / DW_TT;
    float* rp = rowpart + (size_t)row * ntile * 8;
    for (int i = threadIdx.x; i < ntile * 8; i += 256)
        rp[i] = i < 8 ? (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]) : 0.f;
    if (bacc1 && threadIdx.x == 0) {     // gLN1's gamma-weighted totals; the sample's last row publishes the two means (gln_bwd_publish)
        const float g1c = gamma1[c];
        // slots are taken by the global row number; a sample's rows are C consecutive numbers starting at b * C
        const int slot = row & (SEP_STATS_SLOTS - 1), first = b * C;
        int in_slot = 0, used = 0;
        for (int q = 0; q < SEP_STATS_SLOTS; ++q) {
            const int f0 = first + ((q - first) & (SEP_STATS_SLOTS - 1));
            const int cnt_q = f0 < first + C ? (first + C - f0 + SEP_STATS_SLOTS - 1) / SEP_STATS_SLOTS : 0;
            used += cnt_q > 0;
            if (q == slot) in_slot = cnt_q;
        }
        const double t0_ = (double)(g1c * ((part[0][0] + part[1][0]) + (part[2][0] + part[3][0])));
        const double t1_ = (double)(g1c * ((part[0][1] + part[1][1]) + (part[2][1] + part[3][1])));
        if (arrive1)
            gln_bwd_publish(bacc1 + (size_t)b * SEP_STATS_SLOTS * 2, stats1 + (size_t)b * SEP_STATS_SLOTS * 2, arrive1 + (size_t)b * SEP_ARRIVE_INTS, bsum1 + 2 * b,
                            slot, in_slot, used, t0_, t1_, (double)C * T, eps);
        else {          // sums only: the consumer forms the means (sep_gemm_desc.pro_bacc)
            double* ba = bacc1 + ((size_t)b * SEP_STATS_SLOTS + slot) * 2;
            atomicAdd(ba, t0_); atomicAdd(ba + 1, t1_);
        }
    }
}

// =====================================================================================
// gLN backward, second stage, in two small kernels.
//  rows:   one WAVE per (b, c) row reduces that row's per-tile partials (nq in {2, 8}) and writes
//          pbeta[b][c] = R1, pgamma[b][c] = r_b (R2 - mu_b R1), nq == 8: pextra[b] = [db[C] | dw[C][3]] and
//          scratch[b][c] = rowpart[..][6] total (PReLU slope partial)
//  sample: one block per sample sums over channels: bsum[b] = {sum_c gamma_c R1, sum_c gamma_c pgamma} / count,
//          palpha[b] = sum_c scratch[b][c]
// =====================================================================================
__global__ __launch_bounds__(256) void gln_bwd_finalize_rows_kernel(const float* __restrict__ rowpart, int ntile, int nq,
                                                                    const double* __restrict__ stats, double count, float eps,
                                                                    float* __restrict__ pbeta, float* __restrict__ pgamma,
                                                                    float* __restrict__ pextra, float* __restrict__ scratch,
                                                                    int B, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // b*C + c
    if (row >= (long)B * C) return;
    const int b = (int)(row / C), c = (int)(row % C);
    float mu, rstd;
    gln_mu_rstd(stats + (size_t)b * SEP_STATS_SLOTS * 2, count, eps, mu, rstd);
    const int rowlen = ntile * nq;
    const float* rp = rowpart + (size_t)row * rowlen;
    float acc = 0.f;
    for (int i = lane; i < rowlen; i += 64) acc += rp[i];            // i % nq == lane % nq (nq divides 64)
    for (int o = nq; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    // every lane now holds the total of quantity q = lane % nq
    const float R1 = __shfl(acc, 0, 64), R2 = __shfl(acc, 1, 64);
    if (lane == 0) {
        pbeta[row] = R1;
        pgamma[row] = rstd * (R2 - mu * R1);
    }
    if (nq == 8) {
        // per-sample slab of 4C floats: [ db[C] | dw[C][3] ]
        if (lane == 2) pextra[(size_t)b * 4 * C + c] = acc;
        if (lane >= 3 && lane < 6) pextra[(size_t)b * 4 * C + C + (size_t)c * 3 + (lane - 3)] = acc;
        if (lane == 6) scratch[row] = acc;
    }
}

__global__ __launch_bounds__(256) void gln_bwd_finalize_sample_kernel(const float* __restrict__ pbeta, const float* __restrict__ pgamma,
                                                                      const float* __restrict__ gamma, const float* __restrict__ scratch,
                                                                      double count, float* __restrict__ bsum,
                                                                      float* __restrict__ palpha, int C) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    double sg = 0.0, sgx = 0.0, sal = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float gc = gamma[c];
        sg += (double)(gc * pbeta[(size_t)b * C + c]);
        sgx += (double)(gc * pgamma[(size_t)b * C + c]);
        if (palpha) sal += (double)scratch[(size_t)b * C + c];
    }
    const double tg = block_sum_256<double>(sg, red);
    const double tgx = block_sum_256<double>(sgx, red);
    const double tal = block_sum_256<double>(sal, red);
    if (threadIdx.x == 0) {
        if (bsum) {
            bsum[2 * b] = (float)(tg / count);
            bsum[2 * b + 1] = (float)(tgx / count);
        }
        if (palpha) palpha[b] = (float)tal;
    }
}

// The same two stages for up to 64 gLNs in ONE launch each (the Conv-TasNet step queues the second stages of all its layers -- leaves of
// the backward pass -- and flushes them together: 2 launches instead of 2 per layer).
constexpr int FMAXSEG = 64;
struct FinalizeArgs {
    sep_finalize_seg seg[FMAXSEG];
    int blk_start[FMAXSEG + 1];
    int nseg;
};
// lanes that share one row of partials: rowlen / 4 (a float4 each) when that is a power of two <= 64 -- 8 for the depthwise kernel's
// 4 tiles x 8 sums, so a wave reduces EIGHT rows -- else the whole wave walks the row by scalars (the stand-alone kernel's form).  One wave
// per 128-byte row made this launch 133 us for 49 segments (100,000 workgroups of a few instructions each).
__host__ __device__ inline int fin_lanes_per_row(int rowlen, int nq) {
    const int l = rowlen / 4;
    const bool pow2 = rowlen % 4 == 0 && l >= 1 && l <= 64 && (l & (l - 1)) == 0;
    return pow2 && (nq == 2 || l >= 2) ? l : 64;
}
__global__ __launch_bounds__(256) void gln_bwd_finalize_rows_batch_kernel(const FinalizeArgs a) {
    int sgi = 0;
    while (sgi + 1 < a.nseg && (int)blockIdx.x >= a.blk_start[sgi + 1]) ++sgi;
    const sep_finalize_seg sg = a.seg[sgi];
    const int lane = threadIdx.x & 63;
    const int rowlen = sg.ntile * sg.nq;
    const int lpr = fin_lanes_per_row(rowlen, sg.nq), rpw = 64 / lpr;
    const bool packed = lpr != 64 || rowlen == 256;
    const long nrows = (long)sg.B * sg.C;
    const long row = ((long)((int)blockIdx.x - a.blk_start[sgi]) * 4 + (threadIdx.x >> 6)) * rpw + lane / lpr;      // b*C + c
    const int p = lane % lpr;
    const bool live = row < nrows;
    const float* rp = sg.rowpart + (size_t)(live ? row : 0) * rowlen;
    float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // the row's totals of quantity 0 .. nq-1, valid in the row's lane p == 0
    if (packed) {
        float4 v = live ? ld4(rp + 4 * p) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (sg.nq == 8) {                                        // float4 p holds quantities 4 (p & 1) .. + 3
            for (int o = 2; o < lpr; o <<= 1) {
                v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64); v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
            }
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
            q[4] = __shfl_xor(v.x, 1, 64); q[5] = __shfl_xor(v.y, 1, 64); q[6] = __shfl_xor(v.z, 1, 64); q[7] = __shfl_xor(v.w, 1, 64);
        } else {                                                 // nq == 2: elements alternate between the two quantities
            float s0 = v.x + v.z, s1 = v.y + v.w;
            for (int o = 1; o < lpr; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
            q[0] = s0; q[1] = s1;
        }
    } else {
        float acc = 0.f;
        if (live)
            for (int i = lane; i < rowlen; i += 64) acc += rp[i];    // i % nq == lane % nq (nq divides 64)
        for (int o = sg.nq; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = __shfl(acc, k, 64);
    }
    if (!live || p != 0) return;
    const int b = (int)(row / sg.C), c = (int)(row % sg.C);
    float mu, rstd;
    gln_mu_rstd(sg.stats + (size_t)b * SEP_STATS_SLOTS * 2, sg.count, sg.eps, mu, rstd);
    sg.pbeta[row] = q[0];
    sg.pgamma[row] = rstd * (q[1] - mu * q[0]);
    if (sg.nq == 8) {
        float* scratch = sg.pextra + (size_t)sg.B * sg.C * 4 + sg.B;
        sg.pextra[(size_t)b * 4 * sg.C + c] = q[2];
        sg.pextra[(size_t)b * 4 * sg.C + sg.C + (size_t)c * 3 + 0] = q[3];
        sg.pextra[(size_t)b * 4 * sg.C + sg.C + (size_t)c * 3 + 1] = q[4];
        sg.pextra[(size_t)b * 4 * sg.C + sg.C + (size_t)c * 3 + 2] = q[5];
        scratch[row] = q[6];
    }
}
__global__ __launch_bounds__(256) void gln_bwd_finalize_sample_batch_kernel(const FinalizeArgs a) {
    __shared__ double red[4];
    int sgi = 0;
    while (sgi + 1 < a.nseg && (int)blockIdx.x >= a.blk_start[sgi + 1]) ++sgi;      // (blk_start counts samples here)
    const sep_finalize_seg sg = a.seg[sgi];
    const int b = (int)blockIdx.x - a.blk_start[sgi];
    const bool slopes = sg.nq == 8;
    const float* scratch = slopes ? sg.pextra + (size_t)sg.B * sg.C * 4 + sg.B : nullptr;
    double sg1 = 0.0, sgx = 0.0, sal = 0.0;
    for (int c = threadIdx.x; c < sg.C; c += 256) {
        const float gc = sg.gamma[c];
        sg1 += (double)(gc * sg.pbeta[(size_t)b * sg.C + c]);
        sgx += (double)(gc * sg.pgamma[(size_t)b * sg.C + c]);
        if (slopes) sal += (double)scratch[(size_t)b * sg.C + c];
    }
    const double tg = block_sum_256<double>(sg1, red);
    const double tgx = block_sum_256<double>(sgx, red);
    const double tal = block_sum_256<double>(sal, red);
    if (threadIdx.x == 0) {
        if (sg.bsum) {
            sg.bsum[2 * b] = (float)(tg / sg.count);
            sg.bsum[2 * b + 1] = (float)(tgx / sg.count);
        }
        if (slopes) sg.pextra[(size_t)sg.B * sg.C * 4 + b] = (float)tal;
    }
}

// =====================================================================================
// gLN backward statistics FROM the weight gradient.  For a product y = W v (1x1 convolution) behind v = gLN(u), u = PReLU(z), the
// gradient arriving at v is dv = W^T g, and the two row sums the gLN backward needs are contractions the weight gradient has
// already done:
//     R1[n] = sum_t dv[n][t]        = sum_m W[m][n] * (sum_t g[m][t])            = sum_m W[m][n] * gs[m]
//     R2[n] = sum_t dv[n][t] u[n][t] = sum_m W[m][n] * (sum_t g[m][t] u[n][t])    = sum_m W[m][n] * raw[m][n]
// with raw = the weight gradient taken against u instead of v (sep_pw_wgrad with x_mode = PRELU) and gs the bias gradient, both PER
// SAMPLE (sample-aligned slabs).  So the input-gradient product dv = W^T g needs no row-sum epilogue and does not read z at all (one
// H-tensor less per layer and step), and no second-stage kernel runs for this gLN.  The gain / shift of the normalisation, which
// the raw gradient skipped, is applied here on the small matrices:  dW_b[m][n] = sc_bn raw_b[m][n] + sh_bn gs_b[m].
// grid (ceil(N / 32), B), 1024 threads: lane & 31 = column n, (wave, lane >> 5) = 32 row classes -- a thread's rows are m = cls + 32 i, all of
// whose loads are in flight together (the first version walked 64 rows one by one in a 256-thread workgroup: 87 us per launch for 42 MB,
// profiles/r03c_kernel_stats.md).  The sample's last workgroup publishes the two means the consumer of dv reads (gln_bwd_publish).
// =====================================================================================
constexpr int GW_ROWS = 8;        // rows per thread and pass
__global__ __launch_bounds__(1024) void gln_bwd_from_wgrad_kernel(const float* __restrict__ part, const float* __restrict__ part_bias,
                                                                  const float* __restrict__ W, const double* __restrict__ stats,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  double count, float eps, float* __restrict__ dW_b,
                                                                  float* __restrict__ pbeta, float* __restrict__ pgamma,
                                                                  double* __restrict__ bacc, int* __restrict__ arrive, float* __restrict__ bsum,
                                                                  int M, int N, int sps, int accumulate, int products) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // gs[M] | red[32 classes][32 columns][2]
    float* gs = lds;
    float* red = lds + M;
    const int b = blockIdx.y;
    const int col = threadIdx.x & 31, cls = threadIdx.x >> 5;        // 32 row classes
    const int n = blockIdx.x * 32 + col;
    const bool live = n < N;
    for (int m = threadIdx.x; m < M; m += 1024) {
        float t = 0.f;
        for (int k = 0; k < sps; ++k) t += part_bias[(size_t)(b * sps + k) * M + m];
        gs[m] = t;
    }
    float mu, rstd;
    gln_mu_rstd(stats + (size_t)b * SEP_STATS_SLOTS * 2, count, eps, mu, rstd);
    const float gm = live ? gamma[n] : 0.f;
    const float sc = gm * rstd, sh = live ? beta[n] - mu * sc : 0.f;
    __syncthreads();
    float r1 = 0.f, r2 = 0.f;
    const size_t slab = (size_t)M * N;
    const int nc = live ? n : 0;
    const float* p0 = part + (size_t)b * sps * slab + nc;
    for (int m0 = cls; m0 < M; m0 += 32 * GW_ROWS) {
        float raw[GW_ROWS], w[GW_ROWS];
#pragma unroll
        for (int i = 0; i < GW_ROWS; ++i) {
            const int m = m0 + 32 * i;
            const int mc = m < M ? m : 0;                      // rows past M: a harmless in-bounds load, zero weight
            w[i] = (m < M && live) ? W[(size_t)mc * N + nc] : 0.f;
            raw[i] = p0[(size_t)mc * N];
        }
        for (int k = 1; k < sps; ++k)
#pragma unroll
            for (int i = 0; i < GW_ROWS; ++i) {
                const int m = m0 + 32 * i;
                raw[i] += p0[k * slab + (size_t)(m < M ? m : 0) * N];
            }
#pragma unroll
        for (int i = 0; i < GW_ROWS; ++i) {
            const int m = m0 + 32 * i;
            if (m < M && live) {
                const float g = gs[m];
                r1 = fmaf(w[i], g, r1);
                r2 = fmaf(w[i], raw[i], r2);
                dW_b[((size_t)b * M + m) * N + n] = fmaf(sc, raw[i], sh * g);
            }
        }
    }
    red[(cls * 32 + col) * 2] = r1;
    red[(cls * 32 + col) * 2 + 1] = r2;
    __syncthreads();
    if (threadIdx.x < 64) {                                // lanes 0-31: R1 of the block's columns, lanes 32-63: R2
        const int q = threadIdx.x >> 5;
        float R = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) R += red[(k * 32 + col) * 2 + q];
        const float R1 = __shfl(R, col, 64), R2 = __shfl(R, 32 + col, 64);
        if (live && q == 0) {
            const float pg = rstd * (R2 - mu * R1);
            if (accumulate) { pbeta[(size_t)b * N + n] += R1; pgamma[(size_t)b * N + n] += pg; }
            else { pbeta[(size_t)b * N + n] = R1; pgamma[(size_t)b * N + n] = pg; }
        }
        float a = gm * R;                                  // lanes 0-31: gamma R1 ; lanes 32-63: gamma R2
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);      // sums inside each 32-lane half
        const float a2 = __shfl(a, 32, 64);
        if (threadIdx.x == 0) {      // gridDim.x workgroups per sample and product, numbered blockIdx.x: slot blockIdx.x & 15
            const int slot = blockIdx.x & (SEP_STATS_SLOTS - 1);
            gln_bwd_publish(bacc + (size_t)b * SEP_STATS_SLOTS * 2, stats + (size_t)b * SEP_STATS_SLOTS * 2, arrive + (size_t)b * SEP_ARRIVE_INTS, bsum + 2 * b,
                            slot, arrivals_in_slot((int)gridDim.x, slot) * products, slots_in_use((int)gridDim.x), (double)a, (double)a2, count, eps);
        }
    }
}

// dw = r0*(gamma*dvw - mg - xhat*mgx) + dwm  [* (w>0)]   in place on dvw
__global__ __launch_bounds__(256) void head_bwd_kernel(float* __restrict__ dvw, const float* __restrict__ w,
                                                       const float* __restrict__ dwm, const double* __restrict__ stats0,
                                                       const float* __restrict__ gamma0, const float* __restrict__ bsum0,
                                                       int C, int T, int ldt, double count, float eps, int relu) {
    const int row = blockIdx.y;            // b*C + c
    const int b = row / C, c = row % C;
    const int t4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t4 >= ldt) return;
    float mu, rstd;
    gln_mu_rstd(stats0 + (size_t)b * SEP_STATS_SLOTS * 2, count, eps, mu, rstd);
    const float gc = gamma0[c], mg = bsum0[2 * b], mgx = bsum0[2 * b + 1];
    const size_t off = (size_t)row * ldt + t4;
    const float4 g = ld4(dvw + off), ww = ld4(w + off), dm = ld4(dwm + off);
    const float g4[4] = {g.x, g.y, g.z, g.w}, w4[4] = {ww.x, ww.y, ww.z, ww.w}, m4[4] = {dm.x, dm.y, dm.z, dm.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xh = (w4[e] - mu) * rstd;
        float v = rstd * (gc * g4[e] - mg - xh * mgx) + m4[e];
        if (relu && !(w4[e] > 0.f)) v = 0.f;
        o[e] = (t4 + e < T) ? v : 0.f;
    }
    st4(dvw + off, make_float4(o[0], o[1], o[2], o[3]));
}

// =====================================================================================
// Decoder forward: mask*w, basis synthesis, overlap-add, crop.  Thread = frame; loop over basis rows n
// with the decoder row D[n][:] wave-uniform.  LC = Cout*L; frames exchanged through LDS for the OLA.
// Block covers FB = 256-(R-1) output frames, computing R-1 halo frames redundantly (R = L/S).
// =====================================================================================
template <int LC_REG>
__global__ __launch_bounds__(256) void decoder_fwd_kernel(const float* __restrict__ w, const float* __restrict__ m,
                                                          const float* __restrict__ D, float* __restrict__ est,
                                                          float* __restrict__ latent, int n_src, int N, int Cout, int L,
                                                          int S, int F, int ldt, int Tout, int pad_left) {
    extern __shared__ __attribute__((aligned(16))) float ys[];     // [256][LC+1]
    const int LC = Cout * L;
    const int R = L / S;
    // LC_REG > 0: a workgroup owns 64-(R-1) frames and its four waves split the basis rows (2080 workgroups at paper-best;
    // the 256-frame mapping gave 544 = two per CU, each thread a serial loop of 512 dependent load pairs: 1.6 TB/s)
    const int FW = LC_REG > 0 ? 64 : 256;
    const int FB = FW - (R - 1);
    const int bs = blockIdx.y;            // b*n_src + s
    const int b = bs / n_src;
    const int f0 = blockIdx.x * FB;
    const int i = LC_REG > 0 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int wv = threadIdx.x >> 6;
    const int f = f0 - (R - 1) + i;
    const bool fvalid = f >= 0 && f < F;
    const bool owner = i >= R - 1 && f < ldt;       // this block owns latent[f]
    const float* wrow = w + (size_t)b * N * ldt;
    const float* mrow = m + (size_t)bs * N * ldt;
    float* lrow = latent ? latent + (size_t)bs * N * ldt : nullptr;
    const int fc = fvalid ? f : 0;

    if (LC_REG > 0) {
        float acc[LC_REG > 0 ? LC_REG : 1];
#pragma unroll
        for (int q = 0; q < LC_REG; ++q) acc[q] = 0.f;
        const int nper = (N + 3) / 4;
        const int n_lo = wv * nper, n_hi = n_lo + nper < N ? n_lo + nper : N;
        for (int n = n_lo; n < n_hi; ++n) {
            float wh = wrow[(size_t)n * ldt + fc] * mrow[(size_t)n * ldt + fc];
            if (!fvalid) wh = 0.f;
            if (lrow && owner) lrow[(size_t)n * ldt + f] = wh;
            const float* Dn = D + (size_t)n * LC;
#pragma unroll
            for (int q = 0; q < LC_REG; ++q) acc[q] = fmaf(wh, Dn[q], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < LC_REG; ++q) ys[(wv * 64 + i) * (LC + 1) + q] = acc[q];
    } else {
        for (int q0 = 0; q0 < LC; q0 += 16) {
            float acc[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            for (int n = 0; n < N; ++n) {
                float wh = wrow[(size_t)n * ldt + fc] * mrow[(size_t)n * ldt + fc];
                if (!fvalid) wh = 0.f;
                if (q0 == 0 && lrow && owner) lrow[(size_t)n * ldt + f] = wh;
                const float* Dn = D + (size_t)n * LC + q0;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (q0 + q < LC) acc[q] = fmaf(wh, Dn[q], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q0 + q < LC) ys[i * (LC + 1) + q0 + q] = acc[q];
        }
    }
    __syncthreads();
    // overlap-add: padded sample tp = S*fo + k0 gets frames fo - r at tap k0 + r*S
    const int nsamp = FB * S;
    for (int j = threadIdx.x; j < nsamp * Cout; j += 256) {
        const int c = j / nsamp, o = j % nsamp;
        const int fo = o / S, k0 = o % S;            // frame offset inside the block's owned range
        const int tau = S * (f0 + fo) + k0 - pad_left;
        if (tau < 0 || tau >= Tout) continue;
        float v = 0.f;
        for (int r = 0; r < R; ++r) {
            const int cell = (fo + (R - 1) - r) * (LC + 1) + c * L + k0 + r * S;
            if (LC_REG > 0) v += (ys[cell] + ys[64 * (LC + 1) + cell]) + (ys[128 * (LC + 1) + cell] + ys[192 * (LC + 1) + cell]);
            else v += ys[cell];
        }
        est[((size_t)bs * Cout + c) * Tout + tau] = v;
    }
}

// Decoder + mask backward.  Thread = frame, every source handled by the same thread so that dwm = sum_s dlatent*m needs
// no atomics.  raw_mask = 0: dpre = d(sigmoid pre-activation); 1: dpre = d(mask) (softmax masks: sep_softmax_ch_bwd follows).
template <int LC_REG, int NS_REG>
__global__ __launch_bounds__(256) void decoder_bwd_kernel(const float* __restrict__ d_est, const float* __restrict__ w,
                                                          const float* __restrict__ m, const float* __restrict__ D,
                                                          float* __restrict__ dpre, float* __restrict__ dwm, int n_src,
                                                          int N, int Cout, int L, int S, int F, int ldt, int Tout,
                                                          int pad_left, int raw_mask) {
    const int LC = Cout * L;
    const int b = blockIdx.y;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= ldt) return;
    const bool fvalid = f < F;
    const float* wrow = w + (size_t)b * N * ldt;
    float* dwrow = dwm + (size_t)b * N * ldt;
    // basis rows are independent here: blockIdx.z splits them so that the grid fills the chip (16 x B blocks do not)
    const int nper = (N + (int)gridDim.z - 1) / (int)gridDim.z;
    const int n_lo = (int)blockIdx.z * nper;
    const int n_hi = n_lo + nper < N ? n_lo + nper : N;

    auto dsample = [&](int s, int q) -> float {
        const int c = q / L, k = q % L;
        const int tau = S * f + k - pad_left;
        return (fvalid && tau >= 0 && tau < Tout) ? d_est[(((size_t)b * n_src + s) * Cout + c) * Tout + tau] : 0.f;
    };

    if (LC_REG > 0 && NS_REG > 0) {
        float ds[(NS_REG > 0 ? NS_REG : 1)][(LC_REG > 0 ? LC_REG : 1)];
#pragma unroll
        for (int s = 0; s < NS_REG; ++s)
#pragma unroll
            for (int q = 0; q < LC_REG; ++q) ds[s][q] = (s < n_src) ? dsample(s, q) : 0.f;
        for (int n = n_lo; n < n_hi; ++n) {
            const float wv = wrow[(size_t)n * ldt + f];
            const float* Dn = D + (size_t)n * LC;
            float dacc = 0.f;
#pragma unroll
            for (int s = 0; s < NS_REG; ++s) {
                if (s < n_src) {
                    const size_t off = ((size_t)(b * n_src + s) * N + n) * ldt + f;
                    const float mv = m[off];
                    float dl = 0.f;
#pragma unroll
                    for (int q = 0; q < LC_REG; ++q) dl = fmaf(ds[s][q], Dn[q], dl);
                    dpre[off] = fvalid ? (raw_mask ? dl * wv : dl * wv * mv * (1.f - mv)) : 0.f;
                    dacc += dl * mv;
                }
            }
            dwrow[(size_t)n * ldt + f] = fvalid ? dacc : 0.f;
        }
    } else {
        for (int n = n_lo; n < n_hi; ++n) {
            const float wv = wrow[(size_t)n * ldt + f];
            const float* Dn = D + (size_t)n * LC;
            float dacc = 0.f;
            for (int s = 0; s < n_src; ++s) {
                const size_t off = ((size_t)(b * n_src + s) * N + n) * ldt + f;
                const float mv = m[off];
                float dl = 0.f;
                for (int q = 0; q < LC; ++q) dl = fmaf(dsample(s, q), Dn[q], dl);
                dpre[off] = fvalid ? (raw_mask ? dl * wv : dl * wv * mv * (1.f - mv)) : 0.f;
                dacc += dl * mv;
            }
            dwrow[(size_t)n * ldt + f] = fvalid ? dacc : 0.f;
        }
    }
}

// =====================================================================================
// Softmax over the channel axis, in place (reference conv_tasnet.py:353-357, 375: nn.Softmax(dim=1) on the (B, n_src*N, T')
// output of the mask convolution -- over ALL n_src*N channels of a frame, not over the sources).
// Workgroup = 64 frames x 4 channel groups (one wave each); a wave reads 256 contiguous bytes per channel row.
//   forward : y <- exp(y - max_c y) / sum_c exp(y - max_c y)      (frames >= T: 0, the layout contract)
//   backward: g <- y * (g - sum_c g*y)                            (frames >= T: 0)
// =====================================================================================
template <bool BWD>
__global__ __launch_bounds__(256) void softmax_ch_kernel(float* __restrict__ y, float* __restrict__ g, int C, int T, int ldt) {
    __shared__ float sa[4][64], sb[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const size_t base = (size_t)blockIdx.y * C * ldt + t;
    const bool valid = t < T;
    if (!BWD) {
        float mx = -INFINITY, sum = 0.f;
        for (int c = w; c < C; c += 4) {
            const float v = y[base + (size_t)c * ldt];
            const float nm = fmaxf(mx, v);
            sum = sum * expf(mx - nm) + expf(v - nm);
            mx = nm;
        }
        sa[w][lane] = mx; sb[w][lane] = sum;
        __syncthreads();
        float M = fmaxf(fmaxf(sa[0][lane], sa[1][lane]), fmaxf(sa[2][lane], sa[3][lane]));
        float S = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) S += sb[q][lane] * expf(sa[q][lane] - M);
        const float inv = 1.f / S;
        for (int c = w; c < C; c += 4) {
            const size_t off = base + (size_t)c * ldt;
            y[off] = valid ? expf(y[off] - M) * inv : 0.f;
        }
    } else {
        float dot = 0.f;
        for (int c = w; c < C; c += 4) dot = fmaf(g[base + (size_t)c * ldt], y[base + (size_t)c * ldt], dot);
        sa[w][lane] = dot;
        __syncthreads();
        const float tot = (sa[0][lane] + sa[1][lane]) + (sa[2][lane] + sa[3][lane]);
        for (int c = w; c < C; c += 4) {
            const size_t off = base + (size_t)c * ldt;
            g[off] = valid ? y[off] * (g[off] - tot) : 0.f;
        }
    }
}

// =====================================================================================
// Stand-alone gLN (for callers outside the fused network)
// =====================================================================================
__global__ __launch_bounds__(256) void gln_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int C,
                                                        int T, int ldt) {
    __shared__ double red[4];
    const int row = blockIdx.y;
    const int b = row / C;
    // a row is shared by at most 8 workgroups, each thread taking every gridDim.x-th group of 256 float4s: one block reduction and one pair of
    // fp64 atomics per ~8 K frames instead of per 1 K (DPRNN-TasNet's 64,250-frame rows: 8064 workgroups of one float4 per thread took 67 us)
    float s = 0.f, ss = 0.f;
    for (int t4 = (blockIdx.x * 256 + threadIdx.x) * 4; t4 < ldt; t4 += gridDim.x * 1024) {
        const float4 v = ld4(x + (size_t)row * ldt + t4);
        const float v4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t4 + e < T) { s += v4[e]; ss += v4[e] * v4[e]; }
    }
    const double ds = block_sum_256<double>((double)s, red);
    const double dss = block_sum_256<double>((double)ss, red);
    if (threadIdx.x == 0) { double* st = stats + ((size_t)b * SEP_STATS_SLOTS + ((blockIdx.x + blockIdx.y) & (SEP_STATS_SLOTS - 1))) * 2; atomicAdd(st, ds); atomicAdd(st + 1, dss); }
}

__global__ __launch_bounds__(256) void gln_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, int C, int T, int ldt, double count,
                                                        float eps) {
    const int row = blockIdx.y;
    const int b = row / C, c = row % C;
    const int t4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t4 >= ldt) return;
    float mu, rstd;
    gln_mu_rstd(stats + (size_t)b * SEP_STATS_SLOTS * 2, count, eps, mu, rstd);
    const float sc = gamma[c] * rstd, sh = beta[c] - mu * sc;
    const float4 v = ld4(x + (size_t)row * ldt + t4);
    float4 o;
    o.x = (t4 + 0 < T) ? v.x * sc + sh : 0.f;
    o.y = (t4 + 1 < T) ? v.y * sc + sh : 0.f;
    o.z = (t4 + 2 < T) ? v.z * sc + sh : 0.f;
    o.w = (t4 + 3 < T) ? v.w * sc + sh : 0.f;
    st4(y + (size_t)row * ldt + t4, o);
}

// rowpart[b][c][tile][2] = {sum dy, sum dy*x} over a 1024-frame tile (one block per tile)
__global__ __launch_bounds__(256) void gln_bwd_rowsums_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ rowpart, int T, int ldt, int ntile) {
    __shared__ float red[4];
    const int row = blockIdx.y;
    const int tile = blockIdx.x;
    const int t4 = tile * 1024 + threadIdx.x * 4;
    float s1 = 0.f, s2 = 0.f;
    if (t4 < ldt) {
        const float4 g = ld4(dy + (size_t)row * ldt + t4), v = ld4(x + (size_t)row * ldt + t4);
        const float g4[4] = {g.x, g.y, g.z, g.w}, v4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t4 + e < T) { s1 += g4[e]; s2 += g4[e] * v4[e]; }
    }
    const float r1 = block_sum_256<float>(s1, red);
    const float r2 = block_sum_256<float>(s2, red);
    if (threadIdx.x == 0) {
        float* rp = rowpart + ((size_t)row * ntile + tile) * 2;
        rp[0] = r1; rp[1] = r2;
    }
}

__global__ __launch_bounds__(256) void gln_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const double* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ bsum, float* __restrict__ dx, int C,
                                                            int T, int ldt, double count, float eps) {
    const int row = blockIdx.y;
    const int b = row / C, c = row % C;
    const int t4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t4 >= ldt) return;
    float mu, rstd;
    gln_mu_rstd(stats + (size_t)b * SEP_STATS_SLOTS * 2, count, eps, mu, rstd);
    const float gc = gamma[c], mg = bsum[2 * b], mgx = bsum[2 * b + 1];
    const size_t off = (size_t)row * ldt + t4;
    const float4 g = ld4(dy + off), v = ld4(x + off);
    const float g4[4] = {g.x, g.y, g.z, g.w}, v4[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xh = (v4[e] - mu) * rstd;
        o[e] = (t4 + e < T) ? rstd * (gc * g4[e] - mg - xh * mgx) : 0.f;
    }
    st4(dx + off, make_float4(o[0], o[1], o[2], o[3]));
}

__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst,
                                                     int ld_dst, int T) {
    const int row = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ld_dst) return;
    dst[(size_t)row * ld_dst + t] = (t < T) ? src[(size_t)row * ld_src + t] : 0.f;
}


// =====================================================================================
// Segment1d / OverlapAdd1d of the dual-path models (reference src/models/transform.py:6-65) as index maps, without
// the unfold/fold materialisation + permute copies of the reference:
//   segment:     out[b][c][s][k] = xpad[b][c][s*hop + k],  xpad = x (padded rows, T valid frames) shifted right by pad_left
//   overlap-add: out[b][c][t]    = sum_{(s,k): s*hop + k == t + pad_left} y[b][c][s][k]   (t < T; zero for T <= t < ldt)
// The two are adjoints, so each is the other's backward.
// =====================================================================================
__global__ __launch_bounds__(256) void segment_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int ldt,
                                                      int S, int Kc, int hop, int pad_left) {
    const int row = blockIdx.y;                                  // b*C + c
    const int i = blockIdx.x * 256 + threadIdx.x;                // s*Kc + k
    if (i >= S * Kc) return;
    const int s = i / Kc, k = i % Kc;
    const int t = s * hop + k - pad_left;
    out[(size_t)row * S * Kc + i] = (t >= 0 && t < T) ? x[(size_t)row * ldt + t] : 0.f;
}

__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ y, float* __restrict__ out, int T, int ldt,
                                                          int S, int Kc, int hop, int pad_left) {
    const int row = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ldt) return;
    float acc = 0.f;
    if (t < T) {
        const int tp = t + pad_left;
        // chunks covering tp: s*hop <= tp < s*hop + Kc
        int s_hi = tp / hop;
        if (s_hi > S - 1) s_hi = S - 1;
        for (int s = s_hi; s >= 0 && tp - s * hop < Kc; --s) acc += y[((size_t)row * S + s) * Kc + (tp - s * hop)];
    }
    out[(size_t)row * ldt + t] = acc;
}


// =====================================================================================
// Generic depthwise Conv1d (reference src/modules/conv.py:13-29, nn.Conv1d(groups=C) with any kernel size / stride /
// padding / dilation).  Not on the Conv-TasNet hot path (its depthwise is dwconv_fwd above); plain streaming kernels.
//   y[r][to] = bias[c] + sum_k w[c][k] * xpad[r][to*stride + k*dil - pad]           r = b*C + c
// =====================================================================================
__global__ __launch_bounds__(256) void depthwise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int C,
                                                            int Tin, int Tout, int Kw, int stride, int pad, int dil) {
    const int row = blockIdx.y, c = row % C;
    const int to = blockIdx.x * 256 + threadIdx.x;
    if (to >= Tout) return;
    float acc = bias ? bias[c] : 0.f;
    for (int k = 0; k < Kw; ++k) {
        const int ti = to * stride + k * dil - pad;
        if (ti >= 0 && ti < Tin) acc = fmaf(w[c * Kw + k], x[(size_t)row * Tin + ti], acc);
    }
    y[(size_t)row * Tout + to] = acc;
}

__global__ __launch_bounds__(256) void depthwise_bwd_input_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                  float* __restrict__ dx, int C, int Tin, int Tout, int Kw,
                                                                  int stride, int pad, int dil) {
    const int row = blockIdx.y, c = row % C;
    const int ti = blockIdx.x * 256 + threadIdx.x;
    if (ti >= Tin) return;
    float acc = 0.f;
    for (int k = 0; k < Kw; ++k) {
        const int num = ti + pad - k * dil;
        if (num >= 0 && num % stride == 0) {
            const int to = num / stride;
            if (to < Tout) acc = fmaf(w[c * Kw + k], dy[(size_t)row * Tout + to], acc);
        }
    }
    dx[(size_t)row * Tin + ti] = acc;
}

// partial[b][c][0..Kw-1] = sum_to dy * xpad(tap k) ; partial[b][c][Kw] = sum_to dy     (one block per (b, c) row)
__global__ __launch_bounds__(256) void depthwise_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   float* __restrict__ partial, int C, int Tin, int Tout, int Kw,
                                                                   int stride, int pad, int dil) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    for (int k = 0; k <= Kw; ++k) {
        float acc = 0.f;
        for (int to = threadIdx.x; to < Tout; to += 256) {
            const float g = dy[(size_t)row * Tout + to];
            if (k == Kw) acc += g;
            else {
                const int ti = to * stride + k * dil - pad;
                if (ti >= 0 && ti < Tin) acc = fmaf(g, x[(size_t)row * Tin + ti], acc);
            }
        }
        const float tot = block_sum_256<float>(acc, red);
        if (threadIdx.x == 0) partial[(size_t)row * (Kw + 1) + k] = tot;
    }
}

// =====================================================================================
// Global layer norm on TOKEN-MAJOR rows: x (nseq, L, C) with the features contiguous -- the layout the dual-path separators keep between
// their attention / LSTM / Linear layers (DPTNet's and GALRNet's blocks: reference src/models/dptnet.py:505-560, `norm1d(x.permute(1, 2, 0))`).
// Statistics over all L*C values of a sequence (nn.GroupNorm(1, C) on (batch, C, L): mean, biased variance, eps inside the root), gain and
// shift per feature.  One workgroup per sequence, float4 per thread and trip; the second pass over the sequence's <= a few hundred KB
// comes out of L2.  With the features innermost the channel-major sep_gln_* kernels would need two transposing copies per norm
// (DPTNet: 96 of its 244 strided copies per step).  C must divide 1024 (a thread's four lanes keep their features across trips).
//   forward : y = (x - mu) rstd gamma_c + beta_c ; stats[s] = {mu, rstd}
//   backward: dx = rstd (g gamma_c - m1 - xhat m2), m1 = mean(g gamma), m2 = mean(g gamma xhat) ; part[s] = {sum_t g xhat [C] | sum_t g [C]}
// =====================================================================================
__global__ __launch_bounds__(256) void gln_tokens_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, float* __restrict__ stats, int n, int C, float eps) {
    __shared__ double red[4];
    const float* xs = x + (size_t)blockIdx.x * n;
    float* ys = y + (size_t)blockIdx.x * n;
    double s = 0.0, ss = 0.0;
    for (int i = 4 * threadIdx.x; i < n; i += 1024) {
        const float4 v = ld4(xs + i);
        s += (double)((v.x + v.y) + (v.z + v.w));
        ss += (double)(fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w));
    }
    const double ts = block_sum_256<double>(s, red), tss = block_sum_256<double>(ss, red);
    const double m = ts / n;
    double var = tss / n - m * m;
    if (var < 0.0) var = 0.0;
    const float mu = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mu; stats[2 * blockIdx.x + 1] = rstd; }
    const int c0 = (4 * threadIdx.x) % C;
    const float4 g4 = ld4(gamma + c0), b4 = ld4(beta + c0);
    for (int i = 4 * threadIdx.x; i < n; i += 1024) {
        const float4 v = ld4(xs + i);
        st4(ys + i, make_float4(fmaf((v.x - mu) * rstd, g4.x, b4.x), fmaf((v.y - mu) * rstd, g4.y, b4.y),
                                fmaf((v.z - mu) * rstd, g4.z, b4.z), fmaf((v.w - mu) * rstd, g4.w, b4.w)));
    }
}

__global__ __launch_bounds__(256) void gln_tokens_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, float* __restrict__ dx, float* __restrict__ part,
                                                             int n, int C) {
    __shared__ double red[4];
    __shared__ float pc[256][8];
    const float* xs = x + (size_t)blockIdx.x * n;
    const float* gs = dy + (size_t)blockIdx.x * n;
    float* ds = dx + (size_t)blockIdx.x * n;
    const float mu = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    const int c0 = (4 * threadIdx.x) % C;
    const float4 g4 = ld4(gamma + c0);
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
    double s1 = 0.0, s2 = 0.0;
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 4 * threadIdx.x; i < n; i += 1024) {
        const float4 v = ld4(xs + i), g = ld4(gs + i);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - mu) * rstd, gg = gv[e] * ga[e];
            a1 += gg; a2 = fmaf(gg, xh, a2);
            pg[e] = fmaf(gv[e], xh, pg[e]); pb[e] += gv[e];
        }
        s1 += (double)a1; s2 += (double)a2;
    }
    const double t1 = block_sum_256<double>(s1, red), t2 = block_sum_256<double>(s2, red);
    const float m1 = (float)(t1 / n), m2 = (float)(t2 / n);
    for (int i = 4 * threadIdx.x; i < n; i += 1024) {
        const float4 v = ld4(xs + i), g = ld4(gs + i);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - mu) * rstd;
            o[e] = rstd * (gv[e] * ga[e] - m1 - xh * m2);
        }
        st4(ds + i, make_float4(o[0], o[1], o[2], o[3]));
    }
    // per-feature partials: thread t holds features (4 t) % C .. + 3; the threads of a feature are C / 4 apart
#pragma unroll
    for (int e = 0; e < 4; ++e) { pc[threadIdx.x][e] = pg[e]; pc[threadIdx.x][4 + e] = pb[e]; }
    __syncthreads();
    float* ps = part + (size_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float tg = 0.f, tb = 0.f;
        for (int t = c / 4; t < 256; t += C / 4) { tg += pc[t][c & 3]; tb += pc[t][4 + (c & 3)]; }
        ps[c] = tg; ps[C + c] = tb;
    }
}

// ---- few, long sequences (GALRNet: one sequence per SAMPLE, ~1 MB each): one workgroup per sequence leaves the chip to nseq compute
// units (96 / 128 us forward / backward with 4 samples, profiles/r05zj_galrnet_kernel_stats.md).  Split form: grid (nsplit, nseq), slices of
// whole 1024-float trips (a thread keeps its four features), the slices' partial sums in a workspace, every workgroup adding them up in
// slice order for itself; two launches each way.
//   wsd[(s * nsplit + j) * 2 + {0, 1}]            forward: sum, sum of squares of slice j | backward: sum g gamma, sum g gamma xhat
//   wsf[((s * nsplit + j) * 2 + {0, 1}) * C + c]  backward: slice j's sum_t g xhat, sum_t g of feature c
__device__ __forceinline__ void gln_tokens_slice(const int n, const int nsplit, int& lo, int& hi) {
    const int trips = (n + 1023) / 1024, per = (trips + nsplit - 1) / nsplit;
    lo = blockIdx.x * per * 1024;
    hi = lo + per * 1024;
    if (hi > n) hi = n;
}
__global__ __launch_bounds__(256) void gln_tokens_part_fwd_kernel(const float* __restrict__ x, double* __restrict__ wsd, int n, int nsplit) {
    __shared__ double red[4];
    const float* xs = x + (size_t)blockIdx.y * n;
    int lo, hi;
    gln_tokens_slice(n, nsplit, lo, hi);
    double s = 0.0, ss = 0.0;
    for (int i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
        const float4 v = ld4(xs + i);
        s += (double)((v.x + v.y) + (v.z + v.w));
        ss += (double)(fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w));
    }
    const double ts = block_sum_256<double>(s, red), tss = block_sum_256<double>(ss, red);
    if (threadIdx.x == 0) {
        wsd[((size_t)blockIdx.y * nsplit + blockIdx.x) * 2] = ts;
        wsd[((size_t)blockIdx.y * nsplit + blockIdx.x) * 2 + 1] = tss;
    }
}
__global__ __launch_bounds__(256) void gln_tokens_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const double* __restrict__ wsd, float* __restrict__ y, float* __restrict__ stats, int n,
                                                                   int nsplit, int C, float eps) {
    const float* xs = x + (size_t)blockIdx.y * n;
    float* ys = y + (size_t)blockIdx.y * n;
    double ts = 0.0, tss = 0.0;
    for (int j = 0; j < nsplit; ++j) {
        ts += wsd[((size_t)blockIdx.y * nsplit + j) * 2];
        tss += wsd[((size_t)blockIdx.y * nsplit + j) * 2 + 1];
    }
    const double m = ts / n;
    double var = tss / n - m * m;
    if (var < 0.0) var = 0.0;
    const float mu = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && threadIdx.x == 0) { stats[2 * blockIdx.y] = mu; stats[2 * blockIdx.y + 1] = rstd; }
    int lo, hi;
    gln_tokens_slice(n, nsplit, lo, hi);
    const int c0 = (4 * threadIdx.x) % C;
    const float4 g4 = ld4(gamma + c0), b4 = ld4(beta + c0);
    for (int i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
        const float4 v = ld4(xs + i);
        st4(ys + i, make_float4(fmaf((v.x - mu) * rstd, g4.x, b4.x), fmaf((v.y - mu) * rstd, g4.y, b4.y),
                                fmaf((v.z - mu) * rstd, g4.z, b4.z), fmaf((v.w - mu) * rstd, g4.w, b4.w)));
    }
}
__global__ __launch_bounds__(256) void gln_tokens_part_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ stats, double* __restrict__ wsd, float* __restrict__ wsf, int n,
                                                                  int nsplit, int C) {
    __shared__ double red[4];
    __shared__ float pc[256][8];
    const float* xs = x + (size_t)blockIdx.y * n;
    const float* gs = dy + (size_t)blockIdx.y * n;
    const float mu = stats[2 * blockIdx.y], rstd = stats[2 * blockIdx.y + 1];
    const int c0 = (4 * threadIdx.x) % C;
    const float4 g4 = ld4(gamma + c0);
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
    int lo, hi;
    gln_tokens_slice(n, nsplit, lo, hi);
    double s1 = 0.0, s2 = 0.0;
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
        const float4 v = ld4(xs + i), g = ld4(gs + i);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - mu) * rstd, gg = gv[e] * ga[e];
            a1 += gg; a2 = fmaf(gg, xh, a2);
            pg[e] = fmaf(gv[e], xh, pg[e]); pb[e] += gv[e];
        }
        s1 += (double)a1; s2 += (double)a2;
    }
    const double t1 = block_sum_256<double>(s1, red), t2 = block_sum_256<double>(s2, red);
    const size_t rec = (size_t)blockIdx.y * nsplit + blockIdx.x;
    if (threadIdx.x == 0) { wsd[rec * 2] = t1; wsd[rec * 2 + 1] = t2; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { pc[threadIdx.x][e] = pg[e]; pc[threadIdx.x][4 + e] = pb[e]; }
    __syncthreads();
    float* ps = wsf + rec * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float tg = 0.f, tb = 0.f;
        for (int t = c / 4; t < 256; t += C / 4) { tg += pc[t][c & 3]; tb += pc[t][4 + (c & 3)]; }
        ps[c] = tg; ps[C + c] = tb;
    }
}
__global__ __launch_bounds__(256) void gln_tokens_apply_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ stats, const double* __restrict__ wsd, const float* __restrict__ wsf,
                                                                   float* __restrict__ dx, float* __restrict__ part, int n, int nsplit, int C) {
    const float* xs = x + (size_t)blockIdx.y * n;
    const float* gs = dy + (size_t)blockIdx.y * n;
    float* ds = dx + (size_t)blockIdx.y * n;
    const float mu = stats[2 * blockIdx.y], rstd = stats[2 * blockIdx.y + 1];
    double t1 = 0.0, t2 = 0.0;
    for (int j = 0; j < nsplit; ++j) {
        t1 += wsd[((size_t)blockIdx.y * nsplit + j) * 2];
        t2 += wsd[((size_t)blockIdx.y * nsplit + j) * 2 + 1];
    }
    const float m1 = (float)(t1 / n), m2 = (float)(t2 / n);
    if (blockIdx.x == 0) {                                                       // the per-feature sums of the sequence, slices in order
        float* ps = part + (size_t)blockIdx.y * 2 * C;
        for (int c = threadIdx.x; c < 2 * C; c += 256) {
            float t = 0.f;
            for (int j = 0; j < nsplit; ++j) t += wsf[((size_t)blockIdx.y * nsplit + j) * 2 * C + c];
            ps[c] = t;
        }
    }
    int lo, hi;
    gln_tokens_slice(n, nsplit, lo, hi);
    const int c0 = (4 * threadIdx.x) % C;
    const float4 g4 = ld4(gamma + c0);
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
    for (int i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
        const float4 v = ld4(xs + i), g = ld4(gs + i);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - mu) * rstd;
            o[e] = rstd * (gv[e] * ga[e] - m1 - xh * m2);
        }
        st4(ds + i, make_float4(o[0], o[1], o[2], o[3]));
    }
}
// slices per sequence: none while the sequences alone fill the chip; otherwise ~512 workgroups, at least 16 trips each
static int gln_tokens_nsplit(int nseq, long n) {
    if (nseq >= 128) return 1;
    const long trips = (n + 1023) / 1024;
    long ns = (512 + nseq - 1) / nseq;
    if (ns > trips / 16) ns = trips / 16;
    if (ns > 64) ns = 64;
    return ns < 2 ? 1 : (int)ns;
}

// ---- the TCN layers' own depthwise geometry (stride 1, three taps, Tin = Tout = the workspace stride, `pad` zeros in front: (P - 1) d when
// causal), one (b, c) row per workgroup, a float4 of frames per thread and trip.  Tap k sits at t + k d - pad.  For shifts that are multiples
// of 4 the taps are aligned float4 loads; for the others (d = 1, 2: the first two layers of a block) the row goes through LDS once.
// The generic kernels above take one frame per thread: 130 - 144 us per call at the paper-best sizes (profiles/r05o_causal_kernel_stats.md).
template <int MODE>      // 0: forward  y = b + sum_k w_k x[t + k d - pad];  1: input gradient  dx[t] = sum_k w_k dy[t - k d + pad]
__global__ __launch_bounds__(256) void depthwise3_row_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int C, int ldt, int dil, int pad) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    const int row = blockIdx.x, c = row % C;
    const float* xr = x + (size_t)row * ldt;
    float* yr = y + (size_t)row * ldt;
    const float w0 = w[c * 3 + 0], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
    const float bb = (MODE == 0 && bias) ? bias[c] : 0.f;
    // shifts of the three taps relative to the output frame
    const int s0 = MODE == 0 ? -pad : pad, s1 = MODE == 0 ? dil - pad : pad - dil, s2 = MODE == 0 ? 2 * dil - pad : pad - 2 * dil;
    const bool aligned = ((s0 | s1 | s2) & 3) == 0;
    if (!aligned) {
        for (int t = 4 * threadIdx.x; t < ldt; t += 1024) st4(rowbuf + t, ld4(xr + t));
        __syncthreads();
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 4 * threadIdx.x; t < ldt; t += 1024) {
        float a0[4], a1[4], a2[4];
        if (aligned) {
            const float4 v0 = (t + s0 >= 0 && t + s0 < ldt) ? ld4(xr + t + s0) : zero4;
            const float4 v1 = (t + s1 >= 0 && t + s1 < ldt) ? ld4(xr + t + s1) : zero4;
            const float4 v2 = (t + s2 >= 0 && t + s2 < ldt) ? ld4(xr + t + s2) : zero4;
            a0[0] = v0.x; a0[1] = v0.y; a0[2] = v0.z; a0[3] = v0.w;
            a1[0] = v1.x; a1[1] = v1.y; a1[2] = v1.z; a1[3] = v1.w;
            a2[0] = v2.x; a2[1] = v2.y; a2[2] = v2.z; a2[3] = v2.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i0 = t + e + s0, i1 = t + e + s1, i2 = t + e + s2;
                a0[e] = (i0 >= 0 && i0 < ldt) ? rowbuf[i0] : 0.f;
                a1[e] = (i1 >= 0 && i1 < ldt) ? rowbuf[i1] : 0.f;
                a2[e] = (i2 >= 0 && i2 < ldt) ? rowbuf[i2] : 0.f;
            }
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(w2, a2[e], fmaf(w1, a1[e], fmaf(w0, a0[e], bb)));
        st4(yr + t, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// partial[b][c][k] = sum_t dy[t] x[t + k d - pad] (k < 3), partial[b][c][3] = sum_t dy[t]: one row per workgroup, the x row through LDS
__global__ __launch_bounds__(256) void depthwise3_wgrad_row_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ partial,
                                                                   int ldt, int dil, int pad) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    __shared__ float red[4][4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * ldt;
    const float* gr = dy + (size_t)row * ldt;
    for (int t = 4 * threadIdx.x; t < ldt; t += 1024) st4(rowbuf + t, ld4(xr + t));
    __syncthreads();
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    for (int t = 4 * threadIdx.x; t < ldt; t += 1024) {
        const float4 g4 = ld4(gr + t);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i0 = t + e - pad, i1 = i0 + dil, i2 = i1 + dil;
            q0 = fmaf(g[e], (i0 >= 0 && i0 < ldt) ? rowbuf[i0] : 0.f, q0);
            q1 = fmaf(g[e], (i1 >= 0 && i1 < ldt) ? rowbuf[i1] : 0.f, q1);
            q2 = fmaf(g[e], (i2 >= 0 && i2 < ldt) ? rowbuf[i2] : 0.f, q2);
            q3 += g[e];
        }
    }
    q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2); q3 = wave_sum(q3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = q0; red[wv][1] = q1; red[wv][2] = q2; red[wv][3] = q3; }
    __syncthreads();
    if (threadIdx.x < 4) partial[(size_t)row * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the row kernels take the TCN layers' geometry: stride 1, three taps, rows of a multiple of 4 frames that fit the LDS
static inline bool depthwise3_rows(int Tin, int Tout, int Kw, int stride) {
    static const bool off = getenv("SEPK_DEPTHWISE_ROWS") != nullptr && atoi(getenv("SEPK_DEPTHWISE_ROWS")) == 0;
    return !off && Kw == 3 && stride == 1 && Tin == Tout && Tin % 4 == 0 && (size_t)Tin * sizeof(float) <= 128 * 1024;      // one row of LDS: 128 of gfx950's 160 KB
}

}  // namespace

// ---------------------------------------------------------------------------------------
extern "C" int sep_encoder_fwd(const float* x, const float* E, float* w, double* stats, int B, int Cin, int Tin, int N,
                               int L, int S, int F, int ldt, int pad_left, int relu, sep_stream_t stream) {
    SEP_REQUIRE(x && E && w && stats, "sep_encoder_fwd: null pointer");
    SEP_REQUIRE(B > 0 && Cin > 0 && N > 0 && L > 0 && S > 0 && F > 0 && F <= ldt && ldt % 128 == 0, "sep_encoder_fwd: bad sizes (F=%d ldt=%d)", F, ldt);
    const int span = ENC_FT * S + L - S;
    const size_t smem = (size_t)Cin * span * sizeof(float);
    SEP_REQUIRE(smem <= 150 * 1024, "sep_encoder_fwd: Cin*(128*S+L-S) floats exceed LDS (Cin=%d L=%d S=%d)", Cin, L, S);
    if (Cin == 1 && L == 16 && S == 8 && B <= 65535) {
        hipLaunchKernelGGL(encoder_fwd_l16s8_kernel, dim3(ceil_div(ldt, 256), B), dim3(256), 0, (hipStream_t)stream, x, E, w, stats, Tin, N, F, ldt, pad_left, relu);
        SEP_CHECK_LAUNCH("sep_encoder_fwd");
        return 0;
    }
    dim3 grid(ldt / ENC_FT, B);
    if (Cin * L == 16)
        hipLaunchKernelGGL(encoder_fwd_kernel<16>, grid, dim3(256), smem, (hipStream_t)stream, x, E, w, stats, B, Cin, Tin, N, L, S, F, ldt, pad_left, relu);
    else
        hipLaunchKernelGGL(encoder_fwd_kernel<0>, grid, dim3(256), smem, (hipStream_t)stream, x, E, w, stats, B, Cin, Tin, N, L, S, F, ldt, pad_left, relu);
    SEP_CHECK_LAUNCH("sep_encoder_fwd");
    return 0;
}

extern "C" int sep_unfold(const float* x, float* frames, int Bp, int C, int Tin, int L, int S, int F, int ldt,
                          int pad_left, sep_stream_t stream) {
    SEP_REQUIRE(x && frames && Bp > 0 && C > 0 && L > 0 && S > 0 && F > 0 && F <= ldt, "sep_unfold: bad arguments");
    SEP_REQUIRE(Bp <= 65535 && C * L <= 65535, "sep_unfold: grid too large");
    dim3 grid(ceil_div(ldt, 256), C * L, Bp);
    hipLaunchKernelGGL(unfold_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, frames, C, Tin, L, S, F, ldt, pad_left);
    SEP_CHECK_LAUNCH("sep_unfold");
    return 0;
}

extern "C" int sep_dwconv_fwd(const float* a, const double* stats1, const float* gamma1, const float* beta1,
                              const float* alpha1, const float* wd, const float* bd, const float* alpha2, float* z,
                              double* stats2, int B, int C, int T, int ldt, int dilation, float eps, sep_stream_t stream) {
    SEP_REQUIRE(a && stats1 && gamma1 && beta1 && alpha1 && wd && bd && alpha2 && z && stats2, "sep_dwconv_fwd: null pointer");
    SEP_REQUIRE(B > 0 && C > 0 && T > 0 && ldt % 128 == 0 && ldt >= T, "sep_dwconv_fwd: bad sizes");
    SEP_REQUIRE(dilation >= 1 && dilation <= 4096, "sep_dwconv_fwd: dilation %d out of range [1, 4096]", dilation);
    static const bool force_lds = getenv("SEPK_DWCONV_LDS") != nullptr;
    if (!force_lds && (dilation == 1 || dilation == 2 || dilation % 4 == 0) && (long)B * C <= 0x7fffffffL) {
        // two channels per workgroup: the statistics loads, the two block reductions and the two fp64 atomics are paid once per two
        // rows (0.430 -> 0.385 ms per 8 layers; four rows: 0.42; the same in the backward row kernel: slower, 0.79 -> 0.82)
        const int rows = C % 2 == 0 ? 2 : 1;
        const dim3 grid((unsigned)((long)B * C / rows));
#define SEP_DWF(DM) do { \
            if (rows == 2) hipLaunchKernelGGL((dwconv_fwd_direct_kernel<DM, 2>), grid, dim3(256), 0, (hipStream_t)stream, a, stats1, gamma1, beta1, alpha1, wd, bd, alpha2, z, stats2, C, T, ldt, dilation, eps); \
            else hipLaunchKernelGGL((dwconv_fwd_direct_kernel<DM, 1>), grid, dim3(256), 0, (hipStream_t)stream, a, stats1, gamma1, beta1, alpha1, wd, bd, alpha2, z, stats2, C, T, ldt, dilation, eps); \
        } while (0)
        if (dilation == 1) SEP_DWF(1);
        else if (dilation == 2) SEP_DWF(2);
        else SEP_DWF(0);
#undef SEP_DWF
        SEP_CHECK_LAUNCH("sep_dwconv_fwd");
        return 0;
    }
    const int dpad = (dilation + 3) & ~3;
    const size_t smem = 4 * (size_t)(DW_TT + 2 * dpad) * sizeof(float);
    const long total = (long)B * C * ceil_div(ldt, DW_TT);
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), smem, (hipStream_t)stream, a, stats1, gamma1, beta1, alpha1, wd, bd, alpha2, z, stats2, B, C, T, ldt, dilation, dpad, eps);
    SEP_CHECK_LAUNCH("sep_dwconv_fwd");
    return 0;
}

extern "C" int sep_dwconv_bwd(const float* dv2, const float* z, const float* a, const double* stats1, const float* gamma1,
                              const float* beta1, const float* alpha1, const double* stats2, const float* gamma2,
                              const float* alpha2, const float* bsum2, const float* wd, const float* bd, float* dv1, float* rowpart,
                              double* bacc1, int* arrive1, float* bsum1, int B, int C, int T, int ldt, int dilation, float eps, sep_stream_t stream) {
    SEP_REQUIRE(dv2 && z && a && stats1 && gamma1 && beta1 && alpha1 && stats2 && gamma2 && alpha2 && bsum2 && wd && dv1 && rowpart, "sep_dwconv_bwd: null pointer");
    SEP_REQUIRE((arrive1 == nullptr) == (bsum1 == nullptr) && (bacc1 != nullptr || arrive1 == nullptr), "sep_dwconv_bwd: arrive1 and bsum1 come together and need bacc1");
    SEP_REQUIRE(B > 0 && C > 0 && T > 0 && ldt % 128 == 0 && ldt >= T, "sep_dwconv_bwd: bad sizes");
    SEP_REQUIRE(dilation >= 1 && dilation <= 2048, "sep_dwconv_bwd: dilation %d out of range [1, 2048]", dilation);
    static const bool force_tiles = getenv("SEPK_DWCONV_LDS") != nullptr;
    static const bool no_recompute = getenv("SEPK_DWB_RECOMPUTE") != nullptr && atoi(getenv("SEPK_DWB_RECOMPUTE")) == 0;
    if (!force_tiles && ldt <= 8192 && (long)B * C <= 0x7fffffffL) {        // the row (ldt floats of LDS, ldt / 1024 float4 triples in registers)
        // with the depthwise bias at hand z is formed again from `a` instead of being read (two LDS rows): 3 streams of HBM instead of 4
        const bool recomp = bd != nullptr && !no_recompute;
        const size_t rsmem = (size_t)ldt * sizeof(float);      // (the recomputing form's v1 row and the dz row share it)
        const dim3 grid((unsigned)((long)B * C));
#define SEP_DWB(AL, NIT, RC) hipLaunchKernelGGL((dwconv_bwd_row_kernel<AL, NIT, RC>), grid, dim3(256), rsmem, (hipStream_t)stream, dv2, z, bd, a, stats1, gamma1, beta1, alpha1, stats2, gamma2, alpha2, bsum2, wd, dv1, rowpart, bacc1, arrive1, bsum1, C, T, ldt, dilation, eps)
#define SEP_DWB2(AL, NIT) do { if (recomp) SEP_DWB(AL, NIT, true); else SEP_DWB(AL, NIT, false); } while (0)
        const int dm = dilation % 4 == 0 ? 0 : dilation <= 2 ? dilation : 3;
        if (dm == 0) { if (ldt <= 4096) SEP_DWB2(0, 4); else SEP_DWB2(0, 8); }
        else if (dm == 1) { if (ldt <= 4096) SEP_DWB2(1, 4); else SEP_DWB2(1, 8); }
        else if (dm == 2) { if (ldt <= 4096) SEP_DWB2(2, 4); else SEP_DWB2(2, 8); }
        else { if (ldt <= 4096) SEP_DWB2(3, 4); else SEP_DWB2(3, 8); }
#undef SEP_DWB2
#undef SEP_DWB
        SEP_CHECK_LAUNCH("sep_dwconv_bwd");
        return 0;
    }
    const int dpad = (dilation + 3) & ~3;
    const size_t smem = 4 * 2 * (size_t)(DW_TT + 2 * dpad) * sizeof(float);
    const long total = (long)B * C * ceil_div(ldt, DW_TT);
    hipLaunchKernelGGL(dwconv_bwd_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), smem, (hipStream_t)stream, dv2, z, a, stats1, gamma1, beta1, alpha1, stats2, gamma2, alpha2, bsum2, wd, dv1, rowpart, bacc1, arrive1, bsum1, B, C, T, ldt, dilation, dpad, eps);
    SEP_CHECK_LAUNCH("sep_dwconv_bwd");
    return 0;
}

extern "C" int sep_gln_bwd_finalize(const float* rowpart, int ntile, int nq, const double* stats, const float* gamma,
                                    double count, float eps, float* bsum, float* pbeta, float* pgamma, float* pextra,
                                    int B, int C, sep_stream_t stream) {
    SEP_REQUIRE(rowpart && stats && gamma && pbeta && pgamma, "sep_gln_bwd_finalize: null pointer");
    SEP_REQUIRE(nq == 2 || nq == 8, "sep_gln_bwd_finalize: nq must be 2 or 8 (got %d)", nq);
    SEP_REQUIRE(nq == 2 || pextra, "sep_gln_bwd_finalize: nq == 8 needs pextra");
    // pextra layout for nq == 8: B slabs of 4C floats [db[C] | dw[C][3]], then palpha[B], then B*C floats of scratch
    float* palpha = (nq == 8) ? pextra + (size_t)B * C * 4 : nullptr;
    float* scratch = (nq == 8) ? palpha + B : nullptr;
    const long rows = (long)B * C;
    hipLaunchKernelGGL(gln_bwd_finalize_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rowpart, ntile, nq, stats, count, eps, pbeta, pgamma, pextra, scratch, B, C);
    if (bsum || palpha)      // the per-sample stage: the means (stand-alone gLN backward) and / or the PReLU slope partials
        hipLaunchKernelGGL(gln_bwd_finalize_sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pbeta, pgamma, gamma, scratch, count, bsum, palpha, C);
    SEP_CHECK_LAUNCH("sep_gln_bwd_finalize");
    return 0;
}

extern "C" int sep_gln_bwd_finalize_batch(const sep_finalize_seg* segs, int nseg, sep_stream_t stream) {
    SEP_REQUIRE(segs && nseg >= 1 && nseg <= FMAXSEG, "sep_gln_bwd_finalize_batch: 1..64 segments per launch (got %d)", nseg);
    FinalizeArgs rows, samples;
    int rb = 0, sb = 0;
    bool any_sample = false;
    for (int i = 0; i < nseg; ++i) {
        const sep_finalize_seg& g = segs[i];
        SEP_REQUIRE(g.rowpart && g.stats && g.gamma && g.pbeta && g.pgamma && (g.nq == 2 || (g.nq == 8 && g.pextra)) && g.B > 0 && g.C > 0 && g.ntile > 0,
                    "sep_gln_bwd_finalize_batch: bad segment %d", i);
        rows.seg[i] = g; samples.seg[i] = g;
        rows.blk_start[i] = rb; samples.blk_start[i] = sb;
        const int rows_per_block = 4 * (64 / fin_lanes_per_row(g.ntile * g.nq, g.nq));
        rb += (int)(((long)g.B * g.C + rows_per_block - 1) / rows_per_block);
        sb += (g.bsum || g.nq == 8) ? g.B : 0;           // segments that need no per-sample stage take no blocks of it
        any_sample |= (g.bsum || g.nq == 8);
    }
    rows.blk_start[nseg] = rb; samples.blk_start[nseg] = sb;
    rows.nseg = samples.nseg = nseg;
    hipLaunchKernelGGL(gln_bwd_finalize_rows_batch_kernel, dim3(rb), dim3(256), 0, (hipStream_t)stream, rows);
    if (any_sample) hipLaunchKernelGGL(gln_bwd_finalize_sample_batch_kernel, dim3(sb), dim3(256), 0, (hipStream_t)stream, samples);
    SEP_CHECK_LAUNCH("sep_gln_bwd_finalize_batch");
    return 0;
}

extern "C" int sep_gln_bwd_from_wgrad(const float* part, const float* part_bias, const float* W, const double* stats, const float* gamma,
                                      const float* beta, double count, float eps, float* dW_b, float* pbeta, float* pgamma, double* bacc,
                                      int* arrive, float* bsum, int B, int M, int N, int slabs_per_sample, int accumulate, int products,
                                      sep_stream_t stream) {
    SEP_REQUIRE(part && part_bias && W && stats && gamma && beta && dW_b && pbeta && pgamma && bacc && arrive && bsum, "sep_gln_bwd_from_wgrad: null pointer");
    SEP_REQUIRE(B > 0 && B <= 65535 && M > 0 && M <= 8192 && N > 0 && slabs_per_sample > 0 && products >= 1, "sep_gln_bwd_from_wgrad: bad sizes");
    const size_t smem = ((size_t)M + 32 * 32 * 2) * sizeof(float);
    hipLaunchKernelGGL(gln_bwd_from_wgrad_kernel, dim3(ceil_div(N, 32), B), dim3(1024), smem, (hipStream_t)stream, part, part_bias, W, stats, gamma,
                       beta, count, eps, dW_b, pbeta, pgamma, bacc, arrive, bsum, M, N, slabs_per_sample, accumulate, products);
    SEP_CHECK_LAUNCH("sep_gln_bwd_from_wgrad");
    return 0;
}

extern "C" int sep_head_bwd(float* dvw, const float* w, const float* dwm, const double* stats0, const float* gamma0,
                            const float* bsum0, int B, int C, int T, int ldt, double count, float eps, int relu,
                            sep_stream_t stream) {
    SEP_REQUIRE(dvw && w && dwm && stats0 && gamma0 && bsum0 && ldt % 4 == 0, "sep_head_bwd: bad arguments");
    SEP_REQUIRE((long)B * C <= 65535, "sep_head_bwd: B*C too large for grid.y");
    dim3 grid(ceil_div(ldt, 1024), B * C);
    hipLaunchKernelGGL(head_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dvw, w, dwm, stats0, gamma0, bsum0, C, T, ldt, count, eps, relu);
    SEP_CHECK_LAUNCH("sep_head_bwd");
    return 0;
}

namespace {

// =====================================================================================
// Decoder forward for the mono 16-tap / stride-8 basis (Cout*L = 16, R = 2): one workgroup = 64 frames, a 256-byte ALIGNED
// run of every row, of one sample for ALL its sources, so w is read once and every mask once, two full cache lines per
// wave load (the 63-frame tiles of the general kernel start on odd frames: three lines per 256 bytes, and every source
// re-read w: 797 MB fetched for 402).  No halo frame: the eight samples a tile shares with its neighbour are added with
// atomicAdd onto a zeroed `est` by both of them (two addends: the sum does not depend on the order), the rest are stores.
// Four waves split the basis rows; lane = frame; the decoder row D[n][:] is wave-uniform (scalar loads).
// =====================================================================================
template <int NSRC>
__global__ __launch_bounds__(256) void decoder_fwd16_kernel(const float* __restrict__ w, const float* __restrict__ m, const float* __restrict__ D,
                                                            float* __restrict__ est, float* __restrict__ latent, int n_src, int N, int F,
                                                            int ldt, int Tout, int pad_left) {
    constexpr int RU = 4;                             // rows per trip: RU * (1 + n_src) independent loads in flight per lane (8 rows: 197 us instead of 134)
    __shared__ float ys[4][NSRC][64][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * 64, f = f0 + lane;
    const bool fvalid = f < F;
    const float* wrow = w + (size_t)b * N * ldt + f;
    const float* mrow = m + (size_t)b * n_src * N * ldt + f;
    float* lrow = latent ? latent + (size_t)b * n_src * N * ldt + f : nullptr;
    float acc[NSRC][16];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[s][q] = 0.f;
    const int nper = (N + 3) / 4;
    const int n_lo = wv * nper, n_hi = n_lo + nper < N ? n_lo + nper : N;
    for (int n = n_lo; n < n_hi; n += RU) {
        float wv4[RU], mv4[NSRC][RU];
#pragma unroll
        for (int r = 0; r < RU; ++r) {
            const bool ok = n + r < n_hi;
            wv4[r] = ok ? wrow[(size_t)(n + r) * ldt] : 0.f;
#pragma unroll
            for (int s = 0; s < NSRC; ++s) mv4[s][r] = ok && s < n_src ? mrow[((size_t)s * N + n + r) * ldt] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < RU; ++r) {
            if (n + r >= n_hi) break;
            const float* Dn = D + (size_t)(n + r) * 16;
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                if (s >= n_src) break;
                const float wh = fvalid ? wv4[r] * mv4[s][r] : 0.f;
                if (lrow) lrow[((size_t)s * N + n + r) * ldt] = wh;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[s][q] = fmaf(wh, Dn[q], acc[s][q]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int q = 0; q < 16; ++q) ys[wv][s][lane][q] = acc[s][q];
    __syncthreads();
    // overlap-add: padded sample 8*f + k of this tile, o = 8*(f - f0) + k in [0, 520): frame o/8 at tap k (o < 512) and frame o/8 - 1 at tap k + 8 (o >= 8)
    for (int j = threadIdx.x; j < 520 * n_src; j += 256) {
        const int s = j / 520, o = j % 520;
        const int tau = 8 * f0 + o - pad_left;
        if (tau < 0 || tau >= Tout) continue;
        const int fo = o >> 3, k = o & 7;
        float v = 0.f;
        if (fo < 64) v += (ys[0][s][fo][k] + ys[1][s][fo][k]) + (ys[2][s][fo][k] + ys[3][s][fo][k]);
        if (fo >= 1) v += (ys[0][s][fo - 1][k + 8] + ys[1][s][fo - 1][k + 8]) + (ys[2][s][fo - 1][k + 8] + ys[3][s][fo - 1][k + 8]);
        float* dst = est + ((size_t)b * n_src + s) * Tout + tau;
        if (o < 8 || o >= 512) atomicAdd(dst, v);          // shared with the neighbouring tile
        else *dst = v;
    }
}

}  // namespace

extern "C" int sep_decoder_fwd(const float* w, const float* m, const float* D, float* est, float* latent, int B, int n_src,
                               int N, int Cout, int L, int S, int F, int ldt, int Tout, int pad_left, sep_stream_t stream) {
    SEP_REQUIRE(w && m && D && est, "sep_decoder_fwd: null pointer");
    SEP_REQUIRE(L % S == 0 && L / S <= 64, "sep_decoder_fwd: kernel_size %d must be a multiple of stride %d", L, S);
    const int LC = Cout * L;
    SEP_REQUIRE(LC <= 144, "sep_decoder_fwd: Cout*L=%d too large for the LDS frame buffer", LC);
    SEP_REQUIRE((long)B * n_src <= 65535, "sep_decoder_fwd: B*n_src too large");
    if (LC == 16 && L == 16 && S == 8 && n_src <= 4 && ldt % 64 == 0 && B <= 65535) {
        // est is completed by stores and by two-addend atomic adds at the tile seams: it starts from zero
        SEP_REQUIRE(hipMemsetAsync(est, 0, (size_t)B * n_src * Tout * sizeof(float), (hipStream_t)stream) == hipSuccess, "sep_decoder_fwd: memset failed");
        const dim3 g16(ldt / 64, B);
        if (n_src <= 2) hipLaunchKernelGGL(decoder_fwd16_kernel<2>, g16, dim3(256), 0, (hipStream_t)stream, w, m, D, est, latent, n_src, N, F, ldt, Tout, pad_left);
        else hipLaunchKernelGGL(decoder_fwd16_kernel<4>, g16, dim3(256), 0, (hipStream_t)stream, w, m, D, est, latent, n_src, N, F, ldt, Tout, pad_left);
        SEP_CHECK_LAUNCH("sep_decoder_fwd");
        return 0;
    }
    const int R = L / S, FB = (LC == 16 ? 64 : 256) - (R - 1);
    SEP_REQUIRE(FB > 0, "sep_decoder_fwd: kernel_size / stride too large");
    dim3 grid(ceil_div(ldt + R - 1, FB), B * n_src);
    const size_t smem = (size_t)256 * (LC + 1) * sizeof(float);
    if (LC == 16)
        hipLaunchKernelGGL(decoder_fwd_kernel<16>, grid, dim3(256), smem, (hipStream_t)stream, w, m, D, est, latent, n_src, N, Cout, L, S, F, ldt, Tout, pad_left);
    else
        hipLaunchKernelGGL(decoder_fwd_kernel<0>, grid, dim3(256), smem, (hipStream_t)stream, w, m, D, est, latent, n_src, N, Cout, L, S, F, ldt, Tout, pad_left);
    SEP_CHECK_LAUNCH("sep_decoder_fwd");
    return 0;
}

extern "C" int sep_decoder_bwd(const float* d_est, const float* w, const float* m, const float* D, float* dpre, float* dwm,
                               int B, int n_src, int N, int Cout, int L, int S, int F, int ldt, int Tout, int pad_left,
                               int raw_mask, sep_stream_t stream) {
    SEP_REQUIRE(d_est && w && m && D && dpre && dwm, "sep_decoder_bwd: null pointer");
    SEP_REQUIRE(B <= 65535, "sep_decoder_bwd: B too large");
    const int nz = N >= 64 ? 8 : 1;                               // 2048 workgroups at paper-best instead of 256
    dim3 grid(ceil_div(ldt, 256), B, nz);
    const int LC = Cout * L;
    if (LC == 16 && n_src <= 2)
        hipLaunchKernelGGL((decoder_bwd_kernel<16, 2>), grid, dim3(256), 0, (hipStream_t)stream, d_est, w, m, D, dpre, dwm, n_src, N, Cout, L, S, F, ldt, Tout, pad_left, raw_mask);
    else if (LC == 16 && n_src <= 4)
        hipLaunchKernelGGL((decoder_bwd_kernel<16, 4>), grid, dim3(256), 0, (hipStream_t)stream, d_est, w, m, D, dpre, dwm, n_src, N, Cout, L, S, F, ldt, Tout, pad_left, raw_mask);
    else
        hipLaunchKernelGGL((decoder_bwd_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, d_est, w, m, D, dpre, dwm, n_src, N, Cout, L, S, F, ldt, Tout, pad_left, raw_mask);
    SEP_CHECK_LAUNCH("sep_decoder_bwd");
    return 0;
}

extern "C" int sep_softmax_ch_fwd(float* y, int B, int C, int T, int ldt, sep_stream_t stream) {
    SEP_REQUIRE(y && B > 0 && B <= 65535 && C > 0 && T > 0 && ldt % 64 == 0 && ldt >= T, "sep_softmax_ch_fwd: bad arguments");
    hipLaunchKernelGGL((softmax_ch_kernel<false>), dim3(ldt / 64, B), dim3(256), 0, (hipStream_t)stream, y, (float*)nullptr, C, T, ldt);
    SEP_CHECK_LAUNCH("sep_softmax_ch_fwd");
    return 0;
}

extern "C" int sep_softmax_ch_bwd(const float* y, float* g, int B, int C, int T, int ldt, sep_stream_t stream) {
    SEP_REQUIRE(y && g && B > 0 && B <= 65535 && C > 0 && T > 0 && ldt % 64 == 0 && ldt >= T, "sep_softmax_ch_bwd: bad arguments");
    hipLaunchKernelGGL((softmax_ch_kernel<true>), dim3(ldt / 64, B), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(y), g, C, T, ldt);
    SEP_CHECK_LAUNCH("sep_softmax_ch_bwd");
    return 0;
}

extern "C" size_t sep_gln_tokens_ws_bytes(int nseq, int L, int C) {
    if (nseq <= 0 || L <= 0 || C <= 0) return 0;
    const int ns = gln_tokens_nsplit(nseq, (long)L * C);
    return ns == 1 ? 0 : (size_t)nseq * ns * 2 * (sizeof(double) + (size_t)C * sizeof(float));
}

extern "C" int sep_gln_tokens_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, void* ws, int nseq, int L, int C,
                                  float eps, sep_stream_t stream) {
    SEP_REQUIRE(x && gamma && beta && y && stats && nseq > 0 && L > 0 && C >= 4 && 1024 % C == 0, "sep_gln_tokens_fwd: bad arguments (C must divide 1024, C >= 4)");
    SEP_REQUIRE((long)L * C <= 0x7fffffffL && nseq <= 65535, "sep_gln_tokens_fwd: sequence too long / too many sequences");
    const int ns = gln_tokens_nsplit(nseq, (long)L * C);
    if (ns > 1) {
        SEP_REQUIRE(ws != nullptr, "sep_gln_tokens_fwd: this shape needs the workspace of sep_gln_tokens_ws_bytes");
        hipLaunchKernelGGL(gln_tokens_part_fwd_kernel, dim3(ns, nseq), dim3(256), 0, (hipStream_t)stream, x, (double*)ws, L * C, ns);
        hipLaunchKernelGGL(gln_tokens_apply_fwd_kernel, dim3(ns, nseq), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (const double*)ws, y, stats, L * C, ns, C, eps);
    } else {
        hipLaunchKernelGGL(gln_tokens_fwd_kernel, dim3((unsigned)nseq), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats, L * C, C, eps);
    }
    SEP_CHECK_LAUNCH("sep_gln_tokens_fwd");
    return 0;
}

extern "C" int sep_gln_tokens_bwd(const float* dy, const float* x, const float* gamma, const float* stats, float* dx, float* part, void* ws, int nseq,
                                  int L, int C, sep_stream_t stream) {
    SEP_REQUIRE(dy && x && gamma && stats && dx && part && nseq > 0 && L > 0 && C >= 4 && 1024 % C == 0, "sep_gln_tokens_bwd: bad arguments (C must divide 1024, C >= 4)");
    SEP_REQUIRE((long)L * C <= 0x7fffffffL && nseq <= 65535, "sep_gln_tokens_bwd: sequence too long / too many sequences");
    const int ns = gln_tokens_nsplit(nseq, (long)L * C);
    if (ns > 1) {
        SEP_REQUIRE(ws != nullptr, "sep_gln_tokens_bwd: this shape needs the workspace of sep_gln_tokens_ws_bytes");
        double* wsd = (double*)ws;
        float* wsf = (float*)(wsd + (size_t)nseq * ns * 2);
        hipLaunchKernelGGL(gln_tokens_part_bwd_kernel, dim3(ns, nseq), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, stats, wsd, wsf, L * C, ns, C);
        hipLaunchKernelGGL(gln_tokens_apply_bwd_kernel, dim3(ns, nseq), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, stats, (const double*)wsd, (const float*)wsf, dx, part, L * C, ns, C);
    } else {
        hipLaunchKernelGGL(gln_tokens_bwd_kernel, dim3((unsigned)nseq), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, stats, dx, part, L * C, C);
    }
    SEP_CHECK_LAUNCH("sep_gln_tokens_bwd");
    return 0;
}

extern "C" int sep_gln_stats(const float* x, double* stats, int B, int C, int T, int ldt, sep_stream_t stream) {
    SEP_REQUIRE(x && stats && ldt % 4 == 0 && (long)B * C <= 65535, "sep_gln_stats: bad arguments");
    const int tiles = ceil_div(ldt, 1024);
    dim3 grid(tiles < 8 ? tiles : 8, B * C);
    hipLaunchKernelGGL(gln_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, stats, C, T, ldt);
    SEP_CHECK_LAUNCH("sep_gln_stats");
    return 0;
}

extern "C" int sep_gln_apply(const float* x, const double* stats, const float* gamma, const float* beta, float* y, int B,
                             int C, int T, int ldt, double count, float eps, sep_stream_t stream) {
    SEP_REQUIRE(x && stats && gamma && beta && y && ldt % 4 == 0 && (long)B * C <= 65535, "sep_gln_apply: bad arguments");
    dim3 grid(ceil_div(ldt, 1024), B * C);
    hipLaunchKernelGGL(gln_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, y, C, T, ldt, count, eps);
    SEP_CHECK_LAUNCH("sep_gln_apply");
    return 0;
}

extern "C" int sep_gln_bwd_rowsums(const float* dy, const float* x, float* rowpart, int B, int C, int T, int ldt,
                                   sep_stream_t stream) {
    SEP_REQUIRE(dy && x && rowpart && ldt % 4 == 0 && (long)B * C <= 65535, "sep_gln_bwd_rowsums: bad arguments");
    const int ntile = ceil_div(ldt, 1024);
    dim3 grid(ntile, B * C);
    hipLaunchKernelGGL(gln_bwd_rowsums_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, rowpart, T, ldt, ntile);
    SEP_CHECK_LAUNCH("sep_gln_bwd_rowsums");
    return 0;
}

extern "C" int sep_gln_bwd_apply(const float* dy, const float* x, const double* stats, const float* gamma, const float* bsum,
                                 float* dx, int B, int C, int T, int ldt, double count, float eps, sep_stream_t stream) {
    SEP_REQUIRE(dy && x && stats && gamma && bsum && dx && ldt % 4 == 0 && (long)B * C <= 65535, "sep_gln_bwd_apply: bad arguments");
    dim3 grid(ceil_div(ldt, 1024), B * C);
    hipLaunchKernelGGL(gln_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, bsum, dx, C, T, ldt, count, eps);
    SEP_CHECK_LAUNCH("sep_gln_bwd_apply");
    return 0;
}

extern "C" int sep_repack(const float* src, int ld_src, float* dst, int ld_dst, int rows, int T, sep_stream_t stream) {
    SEP_REQUIRE(src && dst && rows > 0 && T > 0 && ld_src >= T && ld_dst >= T, "sep_repack: bad arguments");
    SEP_REQUIRE(rows <= 65535 * 64, "sep_repack: too many rows");
    // grid.y limited to 65535: fold rows
    int done = 0;
    while (done < rows) {
        const int chunk = (rows - done) > 65535 ? 65535 : (rows - done);
        dim3 grid(ceil_div(ld_dst, 256), chunk);
        hipLaunchKernelGGL(repack_kernel, grid, dim3(256), 0, (hipStream_t)stream, src + (size_t)done * ld_src, ld_src, dst + (size_t)done * ld_dst, ld_dst, T);
        done += chunk;
    }
    SEP_CHECK_LAUNCH("sep_repack");
    return 0;
}

extern "C" int sep_segment(const float* x, float* out, int rows, int T, int ldt, int S, int chunk, int hop, int pad_left,
                           sep_stream_t stream) {
    SEP_REQUIRE(x && out && rows > 0 && T > 0 && ldt >= T && S > 0 && chunk > 0 && hop > 0, "sep_segment: bad arguments");
    int done = 0;
    while (done < rows) {
        const int nr = (rows - done) > 65535 ? 65535 : (rows - done);
        dim3 grid(ceil_div(S * chunk, 256), nr);
        hipLaunchKernelGGL(segment_kernel, grid, dim3(256), 0, (hipStream_t)stream, x + (size_t)done * ldt, out + (size_t)done * S * chunk, T, ldt, S, chunk, hop, pad_left);
        done += nr;
    }
    SEP_CHECK_LAUNCH("sep_segment");
    return 0;
}

extern "C" int sep_overlap_add(const float* y, float* out, int rows, int T, int ldt, int S, int chunk, int hop, int pad_left,
                               sep_stream_t stream) {
    SEP_REQUIRE(y && out && rows > 0 && T > 0 && ldt >= T && S > 0 && chunk > 0 && hop > 0, "sep_overlap_add: bad arguments");
    int done = 0;
    while (done < rows) {
        const int nr = (rows - done) > 65535 ? 65535 : (rows - done);
        dim3 grid(ceil_div(ldt, 256), nr);
        hipLaunchKernelGGL(overlap_add_kernel, grid, dim3(256), 0, (hipStream_t)stream, y + (size_t)done * S * chunk, out + (size_t)done * ldt, T, ldt, S, chunk, hop, pad_left);
        done += nr;
    }
    SEP_CHECK_LAUNCH("sep_overlap_add");
    return 0;
}

extern "C" int sep_depthwise_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int Tin, int Tout,
                                 int Kw, int stride, int pad, int dil, sep_stream_t stream) {
    SEP_REQUIRE(x && w && y && B > 0 && C > 0 && Tin > 0 && Tout > 0 && Kw > 0 && stride > 0 && dil > 0 && pad >= 0, "sep_depthwise_fwd: bad arguments");
    if (depthwise3_rows(Tin, Tout, Kw, stride)) {
        // the LDS row is only touched when a tap's shift is not a multiple of 4 frames (the kernel's `aligned`): none is asked for otherwise,
        // which is every dilation >= 4 of the TCN -- 128 KB per workgroup at the longest row would leave one workgroup per compute unit
        const bool al = ((pad | (dil - pad) | (2 * dil - pad)) & 3) == 0;
        hipLaunchKernelGGL((depthwise3_row_kernel<0>), dim3((unsigned)((long)B * C)), dim3(256), al ? (size_t)0 : (size_t)Tin * sizeof(float), (hipStream_t)stream, x, w, bias, y, C, Tin, dil, pad);
        SEP_CHECK_LAUNCH("sep_depthwise_fwd");
        return 0;
    }
    SEP_REQUIRE((long)B * C <= 65535, "sep_depthwise_fwd: B*C too large");
    hipLaunchKernelGGL(depthwise_fwd_kernel, dim3(ceil_div(Tout, 256), B * C), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, C, Tin, Tout, Kw, stride, pad, dil);
    SEP_CHECK_LAUNCH("sep_depthwise_fwd");
    return 0;
}

extern "C" int sep_depthwise_bwd_input(const float* dy, const float* w, float* dx, int B, int C, int Tin, int Tout, int Kw,
                                       int stride, int pad, int dil, sep_stream_t stream) {
    SEP_REQUIRE(dy && w && dx && B > 0 && C > 0 && stride > 0, "sep_depthwise_bwd_input: bad arguments");
    if (depthwise3_rows(Tin, Tout, Kw, stride)) {
        const bool al = ((pad | (pad - dil) | (pad - 2 * dil)) & 3) == 0;          // (see sep_depthwise_fwd)
        hipLaunchKernelGGL((depthwise3_row_kernel<1>), dim3((unsigned)((long)B * C)), dim3(256), al ? (size_t)0 : (size_t)Tin * sizeof(float), (hipStream_t)stream, dy, w, (const float*)nullptr, dx, C, Tin, dil, pad);
        SEP_CHECK_LAUNCH("sep_depthwise_bwd_input");
        return 0;
    }
    SEP_REQUIRE((long)B * C <= 65535, "sep_depthwise_bwd_input: B*C too large");
    hipLaunchKernelGGL(depthwise_bwd_input_kernel, dim3(ceil_div(Tin, 256), B * C), dim3(256), 0, (hipStream_t)stream, dy, w, dx, C, Tin, Tout, Kw, stride, pad, dil);
    SEP_CHECK_LAUNCH("sep_depthwise_bwd_input");
    return 0;
}

extern "C" int sep_depthwise_bwd_weight(const float* dy, const float* x, float* partial, int B, int C, int Tin, int Tout, int Kw,
                                        int stride, int pad, int dil, sep_stream_t stream) {
    SEP_REQUIRE(dy && x && partial && B > 0 && C > 0, "sep_depthwise_bwd_weight: bad arguments");
    if (depthwise3_rows(Tin, Tout, Kw, stride)) {
        hipLaunchKernelGGL(depthwise3_wgrad_row_kernel, dim3((unsigned)((long)B * C)), dim3(256), (size_t)Tin * sizeof(float), (hipStream_t)stream, dy, x, partial, Tin, dil, pad);
        SEP_CHECK_LAUNCH("sep_depthwise_bwd_weight");
        return 0;
    }
    hipLaunchKernelGGL(depthwise_bwd_weight_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, dy, x, partial, C, Tin, Tout, Kw, stride, pad, dil);
    SEP_CHECK_LAUNCH("sep_depthwise_bwd_weight");
    return 0;
}


// =====================================================================================
// sep_split_rows: rows of an activation tensor split once into the {hi, lo} fp16 operand form of the fp16 weight-gradient kernel's G operand
// (wgrad_pc16.hip, G2_pre): the skip gradient dS is the second G source of all 24 heads' weight gradients of a Conv-TasNet step (reference
// src/models/tdcn.py:173,175: the skip / output pointwise convolutions), so its half of the split arithmetic is done here once per step.
// One workgroup per (sample, row); the row (ldt <= 8192 floats) stays in registers between the maximum and the split.
// =====================================================================================
namespace {
constexpr int SPLIT_MAXK = 64;      // slabs per sample (small batches: one sample is cut into up to 64 slabs so that the weight gradient still fills the chip)
typedef __fp16 split_h2_t __attribute__((ext_vector_type(2)));
template <int NIT>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, unsigned* __restrict__ out, int* __restrict__ exps,
                                                         float* __restrict__ sums, int C, int T, int ldt, int k) {
    __shared__ float red[4][SPLIT_MAXK + 1];
    const int row = blockIdx.x;                       // b * C + r
    const int b = row / C, r = row % C;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nq4 = ldt / 4, per = ldt / k;
    float4 v[NIT];
    float m = 0.f;
    float rs[SPLIT_MAXK];
#pragma unroll
    for (int q = 0; q < SPLIT_MAXK; ++q) rs[q] = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int q = threadIdx.x + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq4) {
            v[i] = ld4(x + (size_t)row * ldt + 4 * q);
            const float e4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            float s4 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float val = 4 * q + e < T ? e4[e] : 0.f;
                m = fmaxf(m, fabsf(val));
                s4 += val;
            }
            const int range = (4 * q) / per;              // per is a multiple of 32: the four frames lie in one range
#pragma unroll
            for (int qq = 0; qq < SPLIT_MAXK; ++qq) rs[qq] += qq == range ? s4 : 0.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
#pragma unroll
    for (int qq = 0; qq < SPLIT_MAXK; ++qq) rs[qq] = wave_sum(rs[qq]);
    if (lane == 0) {
        red[wv][SPLIT_MAXK] = m;
#pragma unroll
        for (int qq = 0; qq < SPLIT_MAXK; ++qq) red[wv][qq] = rs[qq];
    }
    __syncthreads();
    m = fmaxf(fmaxf(red[0][SPLIT_MAXK], red[1][SPLIT_MAXK]), fmaxf(red[2][SPLIT_MAXK], red[3][SPLIT_MAXK]));
    const int ex = m > 0.f ? 13 - __builtin_amdgcn_frexp_expf(m) : 0;      // m = f 2^e', f in [0.5, 1): the maximum lands in [2^12, 2^13)
    if (threadIdx.x == 0) exps[row] = ex;
    if (sums && (int)threadIdx.x < k) sums[((size_t)b * k + threadIdx.x) * C + r] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    unsigned* orow = out + (size_t)row * ldt;          // the row in 32-bit words: 32 words per 32-frame line = 16 words of hi pairs, 16 of lo pairs
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int q = threadIdx.x + 256 * i;
        if (q < nq4) {
            const int t = 4 * q;
            const float e4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            float w4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[e] = t + e < T ? __builtin_ldexpf(e4[e], ex) : 0.f;
            unsigned hi[2], lo[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const split_h2_t hh = __builtin_amdgcn_cvt_pkrtz(w4[2 * h], w4[2 * h + 1]);
                hi[h] = __builtin_bit_cast(unsigned, hh);
                const float l0 = w4[2 * h] - (float)hh.x, l1 = w4[2 * h + 1] - (float)hh.y;      // exact in fp32
                lo[h] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
            }
            unsigned* line = orow + (t / 32) * 32;        // 128 bytes per 32 frames
            const int w0 = (t % 32) / 2;                  // word of the frame pair inside the hi half
            line[w0] = hi[0]; line[w0 + 1] = hi[1];
            line[16 + w0] = lo[0]; line[16 + w0 + 1] = lo[1];
        }
    }
}
}  // namespace

extern "C" int sep_split_rows(const float* x, void* out, int32_t* exps, float* sums, int B, int C, int T, int ldt, int k, sep_stream_t stream) {
    SEP_REQUIRE(x && out && exps && B > 0 && C > 0 && T > 0 && ldt >= T && ldt % 128 == 0 && ldt <= 8192, "sep_split_rows: bad sizes (ldt <= 8192)");
    SEP_REQUIRE(k >= 1 && k <= SPLIT_MAXK && ldt % (32 * k) == 0 && (long)B * C <= 0x7fffffffL, "sep_split_rows: k = %d must divide ldt / 32 (<= %d)", k, SPLIT_MAXK);
    const dim3 grid((unsigned)((long)B * C));
    if (ldt <= 4096) hipLaunchKernelGGL(split_rows_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<unsigned*>(out), exps, sums, C, T, ldt, k);
    else hipLaunchKernelGGL(split_rows_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<unsigned*>(out), exps, sums, C, T, ldt, k);
    SEP_CHECK_LAUNCH("sep_split_rows");
    return 0;
}
