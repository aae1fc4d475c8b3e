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
#include <stdlib.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

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


// =====================================================================================================================================
// ONE pass each way for C <= 512 (round 4).  The three launches above read the tensor twice forward (column sums, apply) and the
// gradient and the tensor twice backward; a workgroup that holds ALL channels of a 32-frame tile in registers (8 waves: a wave instruction
// loads 8 channels x 128 B = whole lines, a thread keeps C / 64 float4) needs each byte once -- what it lacks is the prefix (suffix) of
// the column statistics over the EARLIER (later) tiles of its sample.  Those travel by a decoupled look-back chain (Merrill & Garland's
// single-pass scan): a tile publishes the sum of its own 32 columns at once, looks back over its predecessors' records -- 64 at a
// time, one per lane -- adding sums until it meets a record that already carries an inclusive prefix, then publishes its own
// inclusive prefix.  Tiles take their place in the chain from an atomic ticket, so every predecessor a tile waits for has started, on any
// dispatch order; tickets interleave the samples (ticket % B), so B chains advance side by side.  fp64 for everything that is summed over
// time, as above.  Traffic forward 2 x, backward 3 x the tensor (3 x / 5 x before); the parameter gradients' per-tile partial sums
// (12 bytes per tile and channel) are added per sample by a second small launch, in a fixed order.
// =====================================================================================================================================
constexpr int CH_TW = 32;                     // frames per tile

// Records without fences: a release / acquire pair at agent scope writes back / invalidates the XCD's L2 (it is not coherent with the
// other seven) -- with 131 MB of results in flight through that L2 every publication cost about a microsecond and the chain ran
// serially (profiles/r05ze_*: 144 us forward against 94 for three launches).  Instead every 8-byte word of a record is its own
// relaxed agent-scope atomic (written through to where all XCDs see it) and carries its own validity: the value's bits with the sign
// flipped, after -0.0 has been folded into +0.0, are never zero, and zero -- what the records are cleared to -- means "not yet".
__device__ __forceinline__ unsigned long long chain_enc(const double v) { return __builtin_bit_cast(unsigned long long, v + 0.0) ^ 0x8000000000000000ull; }
__device__ __forceinline__ double chain_dec(const unsigned long long w) { return __builtin_bit_cast(double, w ^ 0x8000000000000000ull); }
__device__ __forceinline__ void chain_publish(unsigned long long* rec, const int pos, const double a, const double q) {
    __hip_atomic_store(rec + 2 * pos, chain_enc(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 2 * pos + 1, chain_enc(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sum over the predecessors of chain position pos: 64 records per trip, one per lane (lane 0: the direct predecessor); a predecessor's
// inclusive record ends the walk, its own sums (agg) continue it
__device__ __forceinline__ void chain_lookback(const unsigned long long* agg, const unsigned long long* incl, const int pos, const int lane, double& ca, double& cq) {
    ca = 0.0;
    cq = 0.0;
    for (int j0 = pos - 1; j0 >= 0; j0 -= 64) {
        const int idx = j0 - lane;
        unsigned long long inc, wa = 0, wq = 0;
        int first;
        bool have_inc;
        while (true) {
            have_inc = true;                                                     // before the chain's head: an empty prefix
            bool have = true;
            if (idx >= 0) {                                                      // all four words in one round trip
                const unsigned long long ia = __hip_atomic_load(incl + 2 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long iq = __hip_atomic_load(incl + 2 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long ga = __hip_atomic_load(agg + 2 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long gq = __hip_atomic_load(agg + 2 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                have_inc = ia != 0 && iq != 0;
                wa = have_inc ? ia : ga;
                wq = have_inc ? iq : gq;
                have = wa != 0 && wq != 0;
            }
            inc = __builtin_amdgcn_ballot_w64(have_inc);
            const unsigned long long ready = __builtin_amdgcn_ballot_w64(have);
            first = inc ? __builtin_ctzll(inc) : 64;
            const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);      // lanes 0 .. first must have published
            if ((ready & need) == need) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const bool use = idx >= 0 && lane <= first;
        ca += wave_sum(use ? chain_dec(wa) : 0.0);
        cq += wave_sum(use ? chain_dec(wq) : 0.0);
        if (inc) break;
    }
}

// R = ceil(C / 64) rounds of 64 channels.  Thread (wave w, lane): channel 64 j + 8 w + (lane >> 3) in round j, frames 4 (lane & 7) .. + 3 of the tile.
template <int R>
__global__ __launch_bounds__(512) void cln_chain_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ alpha, float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                            int* __restrict__ ticket, unsigned long long* __restrict__ agg, unsigned long long* __restrict__ incl,
                                                            int B, int C, int T, int ldt, int nt, float eps) {
    __shared__ float red[2][8][CH_TW];
    __shared__ float mr[2][CH_TW];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x % B;                                                // the samples' chains advance side by side
    if (tid == 0) s_ticket = atomicAdd(ticket + 32 * b, 1);                      // one counter per sample, 128 B apart: a single counter serialises
    __syncthreads();                                                             // every workgroup of the launch at ~70 ns each (2048 tiles: 140 us)
    const int tile = s_ticket;
    const int fl = lane & 7, cl = lane >> 3;
    const int t4 = tile * CH_TW + 4 * fl;
    const bool inrow = t4 < ldt;
    const bool act = alpha != nullptr;
    const float al = act ? alpha[0] : 1.f;
    float4 u[R];
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int c = 64 * j + 8 * w + cl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C && inrow) v = ld4(x + ((size_t)b * C + c) * ldt + t4);
        if (act) { v.x = prelu_f(v.x, al); v.y = prelu_f(v.y, al); v.z = prelu_f(v.z, al); v.w = prelu_f(v.w, al); }
        u[j] = v;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        s1[0] += u[j].x; s1[1] += u[j].y; s1[2] += u[j].z; s1[3] += u[j].w;
        s2[0] = fmaf(u[j].x, u[j].x, s2[0]); s2[1] = fmaf(u[j].y, u[j].y, s2[1]); s2[2] = fmaf(u[j].z, u[j].z, s2[2]); s2[3] = fmaf(u[j].w, u[j].w, s2[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                                                // over the wave's 8 channels: lanes 8, 16, 32 apart
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) { s1[e] += __shfl_xor(s1[e], off, 64); s2[e] += __shfl_xor(s2[e], off, 64); }
    }
    if (cl == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][w][4 * fl + e] = s1[e]; red[1][w][4 * fl + e] = s2[e]; }
    }
    __syncthreads();
    if (w == 0) {
        const int f = lane & 31, t = tile * CH_TW + f;
        double a = 0.0, q = 0.0;
        if (lane < CH_TW && t < T) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { a += (double)red[0][i][f]; q += (double)red[1][i][f]; }
        }
#pragma unroll
        for (int off = 1; off < CH_TW; off <<= 1) {                              // inclusive prefix over the tile's frames (lanes 32 .. 63 carry zeros)
            const double ua = __shfl_up(a, off, 64), uq = __shfl_up(q, off, 64);
            if (lane >= off) { a += ua; q += uq; }
        }
        const double ta = __shfl(a, CH_TW - 1, 64), tq = __shfl(q, CH_TW - 1, 64);      // the tile's sums
        unsigned long long* ab = agg + (size_t)b * nt * 2;
        unsigned long long* ib = incl + (size_t)b * nt * 2;
        double ca = 0.0, cq = 0.0;
        if (tile == 0) {
            if (lane == 0) chain_publish(ib, 0, ta, tq);
        } else {
            if (lane == 0) chain_publish(ab, tile, ta, tq);
            chain_lookback(ab, ib, tile, lane, ca, cq);
            if (lane == 0) chain_publish(ib, tile, ca + ta, cq + tq);
        }
        if (lane < CH_TW) {
            float mf = 0.f, rf = 0.f;
            if (t < T) {
                const double n = (double)C * (double)(t + 1);
                const double m = (ca + a) / n;
                double var = (cq + q) / n - m * m;
                if (var < 0.0) var = 0.0;
                mf = (float)m;
                rf = (float)(1.0 / (sqrt(var) + (double)eps));
                mean[(size_t)b * ldt + t] = mf;
                rstd[(size_t)b * ldt + t] = rf;
            }
            mr[0][f] = mf;
            mr[1][f] = rf;                                                       // 0 for dead frames: they come out as zeros
        }
    }
    __syncthreads();
    if (!inrow) return;
    const float4 m4 = ld4(&mr[0][4 * fl]), r4 = ld4(&mr[1][4 * fl]);
    const bool l0 = t4 < T, l1 = t4 + 1 < T, l2 = t4 + 2 < T, l3 = t4 + 3 < T;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int c = 64 * j + 8 * w + cl;
        if (c >= C) continue;
        const float ga = gamma[c], be = beta[c];
        float4 o;
        o.x = l0 ? (u[j].x - m4.x) * r4.x * ga + be : 0.f;
        o.y = l1 ? (u[j].y - m4.y) * r4.y * ga + be : 0.f;
        o.z = l2 ? (u[j].z - m4.z) * r4.z * ga + be : 0.f;
        o.w = l3 ? (u[j].w - m4.w) * r4.w * ga + be : 0.f;
        st4(y + ((size_t)b * C + c) * ldt + t4, o);
    }
}

// backward: the chain runs from the LAST tile of a sample to the first (suffix sums); parts[(b * nt + tile) * 3 * C + {0, 1, 2} * C + c] =
// the tile's contributions to d(gamma_c), d(beta_c), d(alpha)
// NW waves per workgroup
template <int R, int NW>
__global__ __launch_bounds__(64 * NW) void cln_chain_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ alpha, float* __restrict__ dx, float* __restrict__ parts,
                                                            int* __restrict__ ticket, unsigned long long* __restrict__ agg, unsigned long long* __restrict__ incl,
                                                            int B, int C, int T, int ldt, int nt, float eps) {
    __shared__ float red[2][NW][CH_TW];
    __shared__ float pq[2][CH_TW];
    __shared__ float mrs[2][CH_TW];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x % B;
    if (tid == 0) s_ticket = atomicAdd(ticket + 32 * b, 1);
    __syncthreads();
    const int pos = s_ticket, tile = nt - 1 - pos;
    const int fl = lane & 7, cl = lane >> 3;
    const int t4 = tile * CH_TW + 4 * fl;
    const bool inrow = t4 < ldt;
    const bool act = alpha != nullptr;
    const float al = act ? alpha[0] : 1.f;
    const bool lv[4] = {t4 < T, t4 + 1 < T, t4 + 2 < T, t4 + 3 < T};
    float ms[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (inrow) {
        const float4 m4 = ld4(mean + (size_t)b * ldt + t4), r4 = ld4(rstd + (size_t)b * ldt + t4);
        ms[0] = lv[0] ? m4.x : 0.f; ms[1] = lv[1] ? m4.y : 0.f; ms[2] = lv[2] ? m4.z : 0.f; ms[3] = lv[3] ? m4.w : 0.f;
        rs[0] = lv[0] ? r4.x : 0.f; rs[1] = lv[1] ? r4.y : 0.f; rs[2] = lv[2] ? r4.z : 0.f; rs[3] = lv[3] ? r4.w : 0.f;
    }
    float xv[R][4], gv[R][4];                                                    // x as stored (before the PReLU) and the incoming gradient
    float sA[4] = {0.f, 0.f, 0.f, 0.f}, sB[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int c = 8 * NW * j + 8 * w + cl;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C && inrow) {
            a = ld4(x + ((size_t)b * C + c) * ldt + t4);
            d = ld4(g + ((size_t)b * C + c) * ldt + t4);
        }
        xv[j][0] = a.x; xv[j][1] = a.y; xv[j][2] = a.z; xv[j][3] = a.w;
        gv[j][0] = lv[0] ? d.x : 0.f; gv[j][1] = lv[1] ? d.y : 0.f; gv[j][2] = lv[2] ? d.z : 0.f; gv[j][3] = lv[3] ? d.w : 0.f;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int c = 8 * NW * j + 8 * w + cl;
        const float ga = c < C ? gamma[c] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float uu = act ? prelu_f(xv[j][e], al) : xv[j][e];
            const float gg = gv[j][e] * ga;
            sA[e] += gg;
            sB[e] = fmaf(gg, uu - ms[e], sB[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) { sA[e] += __shfl_xor(sA[e], off, 64); sB[e] += __shfl_xor(sB[e], off, 64); }
    }
    if (cl == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][w][4 * fl + e] = sA[e]; red[1][w][4 * fl + e] = sB[e]; }
        if (w == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { mrs[0][4 * fl + e] = ms[e]; mrs[1][4 * fl + e] = rs[e]; }
        }
    }
    __syncthreads();
    if (w == 0) {
        const int f = lane & 31, t = tile * CH_TW + f;
        double dm = 0.0, dq = 0.0;
        if (lane < CH_TW && t < T) {
            double A = 0.0, Bq = 0.0;
#pragma unroll
            for (int i = 0; i < NW; ++i) { A += (double)red[0][i][f]; Bq += (double)red[1][i][f]; }
            const double r = (double)mrs[1][f], m = (double)mrs[0][f];
            const double sigma = 1.0 / r - (double)eps;
            const double n = (double)C * (double)(t + 1);
            const double Dq = sigma > 0.0 ? -Bq * r * r / (2.0 * sigma) : 0.0;      // as cln_scan_bwd_kernel
            const double Dm = -r * A - 2.0 * m * Dq;
            dm = Dm / n;
            dq = Dq / n;
        }
#pragma unroll
        for (int off = 1; off < CH_TW; off <<= 1) {                              // inclusive SUFFIX over the tile's frames (lanes 32 .. 63 hold zeros)
            const double ua = __shfl_down(dm, off, 64), uq = __shfl_down(dq, off, 64);
            if (lane + off < 64) { dm += ua; dq += uq; }
        }
        const double ta = __shfl(dm, 0, 64), tq = __shfl(dq, 0, 64);
        unsigned long long* ab = agg + (size_t)b * nt * 2;
        unsigned long long* ib = incl + (size_t)b * nt * 2;
        double ca = 0.0, cq = 0.0;
        if (pos == 0) {
            if (lane == 0) chain_publish(ib, 0, ta, tq);
        } else {
            if (lane == 0) chain_publish(ab, pos, ta, tq);
            chain_lookback(ab, ib, pos, lane, ca, cq);
            if (lane == 0) chain_publish(ib, pos, ca + ta, cq + tq);
        }
        if (lane < CH_TW) {
            pq[0][f] = t < T ? (float)(ca + dm) : 0.f;
            pq[1][f] = t < T ? (float)(cq + dq) : 0.f;
        }
    }
    __syncthreads();
    // (the products formed for the column sums -- PReLU(x), g gamma -- are formed AGAIN below: kept alive across the chain they cost 64
    //  registers and an occupancy step; the empty asm hides the equality from the compiler)
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(xv[j][e]), "+v"(gv[j][e]));
    const float4 P4 = ld4(&pq[0][4 * fl]), Q4 = ld4(&pq[1][4 * fl]);
    const float Ps[4] = {P4.x, P4.y, P4.z, P4.w}, Qs[4] = {Q4.x, Q4.y, Q4.z, Q4.w};
    float* prow = parts + (size_t)(b * nt + tile) * 3 * C;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int c = 8 * NW * j + 8 * w + cl;
        const float ga = c < C ? gamma[c] : 0.f;
        float o[4], pg = 0.f, pb = 0.f, pa = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xe = xv[j][e], gl = gv[j][e];
            const float uu = act ? prelu_f(xe, al) : xe;
            const float du = lv[e] ? gl * ga * rs[e] + Ps[e] + 2.f * uu * Qs[e] : 0.f;
            o[e] = act ? du * prelu_grad(xe, al) : du;
            if (act && xe <= 0.f) pa = fmaf(du, xe, pa);
            pg += gl * (uu - ms[e]) * rs[e];
            pb += gl;
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {                                  // over the 8 lanes that hold the row's 32 frames
            pg += __shfl_xor(pg, off, 64);
            pb += __shfl_xor(pb, off, 64);
            if (act) pa += __shfl_xor(pa, off, 64);
        }
        if (c < C) {
            if (inrow) st4(dx + ((size_t)b * C + c) * ldt + t4, make_float4(o[0], o[1], o[2], o[3]));
            if (fl == 0) {
                prow[c] = pg;
                prow[C + c] = pb;
                prow[2 * C + c] = act ? pa : 0.f;
            }
        }
    }
}

// out[b][c] = sum over the tiles of sample b of parts[(b * nt + tile) * 3 * C + k * C + c]   (k = blockIdx.y: gamma | beta | alpha), in a fixed
// order: 16 phases of tiles (tile % 16) per channel, four loads in flight per thread, the phases added in order through LDS
__global__ __launch_bounds__(1024) void cln_chain_parts_kernel(const float* __restrict__ parts, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                               float* __restrict__ dalpha_part, int B, int C, int nt) {
    __shared__ double acc[16][64];
    const int k = blockIdx.y, b = blockIdx.z;
    float* out = k == 0 ? dgamma_part : (k == 1 ? dbeta_part : dalpha_part);
    if (out == nullptr) return;
    const int cc = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + cc;
    double s = 0.0;
    if (c < C) {
        const float* p = parts + (size_t)b * nt * 3 * C + (size_t)k * C + c;
        const size_t st = (size_t)3 * C;
        int t = ph;
        for (; t + 48 < nt; t += 64) {
            const float v0 = p[(size_t)t * st], v1 = p[(size_t)(t + 16) * st], v2 = p[(size_t)(t + 32) * st], v3 = p[(size_t)(t + 48) * st];
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; t < nt; t += 16) s += (double)p[(size_t)t * st];
    }
    acc[ph][cc] = s;
    __syncthreads();
    if (ph == 0 && c < C) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += acc[i][cc];
        out[(size_t)b * C + c] = (float)tot;
    }
}

struct ChainWs {
    int* ticket;
    unsigned long long* agg;
    unsigned long long* incl;
    float* parts;
    size_t bytes, clear_bytes;
};
inline ChainWs chain_ws(double* ws, int B, int C, int nt) {
    ChainWs r;
    char* p = reinterpret_cast<char*>(ws);
    const size_t nrec = (size_t)B * nt;
    const size_t o_agg = 128 * (size_t)B, o_incl = o_agg + 16 * nrec, o_parts = o_incl + 16 * nrec;
    r.ticket = reinterpret_cast<int*>(p);                                        // one counter per sample, 128 B apart
    r.agg = reinterpret_cast<unsigned long long*>(p + o_agg);
    r.incl = reinterpret_cast<unsigned long long*>(p + o_incl);
    r.parts = reinterpret_cast<float*>(p + o_parts);
    r.clear_bytes = o_parts;                                                     // the tickets and the records start every call at zero
    r.bytes = o_parts + nrec * 3 * (size_t)C * 4;
    return r;
}
inline bool chain_takes(int C) { return C <= 512; }

}  // namespace

extern "C" int sep_cln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, double* ws,
                           int B, int C, int T, int ldt, float eps, const float* alpha, sep_stream_t stream_) {
    SEP_REQUIRE(x && gamma && beta && y && mean && rstd && ws && B > 0 && B <= 65535 && C > 0 && T > 0 && ldt >= T && ldt % 4 == 0, "sep_cln_fwd: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    static const bool three = getenv("SEPK_CLN_CHAIN") != nullptr && atoi(getenv("SEPK_CLN_CHAIN")) == 0;      // the three-launch form, for A/B runs
    if (chain_takes(C) && !three) {
        const int nt = ceil_div(ldt, CH_TW);                 // every frame of the rows is written: dead ones as zeros
        SEP_REQUIRE((long)B * nt <= 0x7fffffffL / 4, "sep_cln_fwd: too many tiles");
        const ChainWs cw = chain_ws(ws, B, C, nt);
        SEP_REQUIRE(hipMemsetAsync(ws, 0, cw.clear_bytes, stream) == hipSuccess, "sep_cln: clearing the chain records failed");
#define SEP_CLF(RR) hipLaunchKernelGGL((cln_chain_fwd_kernel<RR>), dim3(B * nt), dim3(512), 0, stream, x, gamma, beta, alpha, y, mean, rstd, cw.ticket, cw.agg, cw.incl, B, C, T, ldt, nt, eps)
        if (C <= 64) SEP_CLF(1); else if (C <= 128) SEP_CLF(2); else if (C <= 256) SEP_CLF(4); else SEP_CLF(8);
#undef SEP_CLF
        SEP_CHECK_LAUNCH("sep_cln_fwd");
        return 0;
    }
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
    static const bool three = getenv("SEPK_CLN_CHAIN") != nullptr && atoi(getenv("SEPK_CLN_CHAIN")) == 0;
    if (chain_takes(C) && !three) {
        const int nt = ceil_div(ldt, CH_TW);                 // every frame of the rows is written: dead ones as zeros
        SEP_REQUIRE((long)B * nt <= 0x7fffffffL / 4, "sep_cln_bwd: too many tiles");
        const ChainWs cw = chain_ws(ws, B, C, nt);
        SEP_REQUIRE(hipMemsetAsync(ws, 0, cw.clear_bytes, stream) == hipSuccess, "sep_cln: clearing the chain records failed");
#define SEP_CLB(RR, NW) hipLaunchKernelGGL((cln_chain_bwd_kernel<RR, NW>), dim3(B * nt), dim3(64 * NW), 0, stream, dy, x, mean, rstd, gamma, alpha, dx, cw.parts, cw.ticket, cw.agg, cw.incl, B, C, T, ldt, nt, eps)
        // C > 256: 8 waves x 8 rounds (124 registers: two workgroups per compute unit, one walks its chain while the other streams) beat
        // 16 waves x 4 rounds (88 registers but one workgroup per CU): 118 against 143 us at B = 16, C = 512 (profiles/r05ze_cln_chain.txt)
        if (C <= 64) SEP_CLB(1, 8); else if (C <= 128) SEP_CLB(2, 8); else if (C <= 256) SEP_CLB(4, 8); else SEP_CLB(8, 8);
#undef SEP_CLB
        hipLaunchKernelGGL(cln_chain_parts_kernel, dim3(ceil_div(C, 64), 3, B), dim3(1024), 0, stream, (const float*)cw.parts, dgamma_part, dbeta_part, dalpha_part, B, C, nt);
        SEP_CHECK_LAUNCH("sep_cln_bwd");
        return 0;
    }
    hipLaunchKernelGGL((cln_colsums_kernel<true>), dim3(ceil_div(T, CLN_TCOLS), B), dim3(256), 0, stream, x, dy, gamma, mean, alpha, ws, C, T, ldt);
    hipLaunchKernelGGL(cln_scan_bwd_kernel, dim3(B), dim3(1024), 0, stream, ws, mean, rstd, C, T, ldt, eps);
    hipLaunchKernelGGL(cln_apply_bwd_kernel, dim3(ceil_div(C, 4), B), dim3(256), 0, stream, dy, x, mean, rstd, (const double*)ws, gamma, alpha, dx, dgamma_part, dbeta_part, dalpha_part, C, T, ldt);
    SEP_CHECK_LAUNCH("sep_cln_bwd");
    return 0;
}

/* bytes of workspace sep_cln_fwd / sep_cln_bwd need for this shape (the chain's records and the per-tile parameter-gradient partials, or the
 * column sums of the three-launch form) */
extern "C" size_t sep_cln_ws_bytes(int B, int C, int T, int ldt) {
    if (B <= 0 || C <= 0 || T <= 0 || ldt < T) return 0;
    size_t need = (size_t)B * 2 * ldt * sizeof(double);
    if (chain_takes(C)) {
        const size_t chain = chain_ws(nullptr, B, C, ceil_div(ldt, CH_TW)).bytes;
        if (chain > need) need = chain;
    }
    return need;
}
