// Scaled dot-product attention of the dual-path separators' transformer blocks, forward and backward, in fp32 on the matrix pipe
// (v_mfma_f32_32x32x2_f32):  O = dropout(softmax(scale Q K^T)) V  per (sequence, head), sequences of at most 320 steps, heads 8 / 16 / 32 wide.
//
// Replaces the core of nn.MultiheadAttention as the reference's blocks call it (src/models/dptnet.py:505-527 MultiheadAttentionBlock,
// src/models/galr.py:160-226, nn.TransformerEncoderLayer of src/models/sepformer.py:395-520) -- torch's memory-efficient kernel for fp32 takes
// 96 us forward and 407 us backward at DPTNet's shape (257 sequences x 4 heads x 250 steps x 16), 188 / 614 us at SepFormer's
// (profiles/r05zn_sdpa_backends.txt), a sixth of DPTNet's and a fifth of SepFormer's step.
//
// Operands stay where the projections leave them: qkv (N, L, 3, H, D) -- the packed input projection's output viewed, no transposes --,
// o (N, L, H, D) = the (N, L, C) rows the output projection reads.  A wave owns 32 queries (forward, dQ) or 32 keys (dK, dV) and walks the
// other index in blocks of 32 through row-major LDS tiles.  The score block is computed TRANSPOSED in the query-major kernels,
// S^T = K Q^T: in the MFMA's result layout a lane then is one QUERY (its registers are keys), so the softmax is lane-local (plus one
// exchange with lane ^ 32, which holds the other half of the keys) and the probabilities are already the A operand of P V -- the k slots
// of an instruction may pair any two contraction indices as long as A and B agree, and B (V's rows) is simply fetched in the order the
// registers imply.  The key-major kernel computes S = Q K^T for the same reason (lane = key): P^T and dS^T are its A operands as they are.
//   forward : S^T (all key blocks in registers: 16 per block), softmax, lse, P V
//   backward: (a) per query block: S^T and dP^T = V dO^T block by block, dS = P o (dP - delta), dQ = scale dS K; writes delta = rowsum(dO o O)
//             (b) per key block: S and dP block by block over the queries, dV = Pd^T dO, dK = scale dS^T Q
// Dropout on the probabilities (nn.TransformerEncoderLayer hands its rate to the attention): a counter-based hash of (seed, n, h, q, key)
// decides each element, the same in all three kernels -- no mask tensor.
#include "common.hpp"

namespace {

typedef float att_f32x16 __attribute__((ext_vector_type(16)));
constexpr int ATT_MAXL = 320;          // (two sizes of every kernel: tiles for 256 and for 320 steps)

__device__ __forceinline__ float4 ald4(const float* p) { return *reinterpret_cast<const float4*>(p); }
constexpr float ATT_LOG2E = 1.4426950408889634f, ATT_LN2 = 0.6931471805599453f;
// the scores are kept in units of log 2 (q, or k, is scaled by scale * log2(e) on load): v_exp_f32 with nothing in front of it
__device__ __forceinline__ float att_exp2(const float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int att_row(const int r, const int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }      // row of result register r
// idx: the element's 64-bit index ((n H + h) L + q) L + key -- beyond 2^32 probabilities (N H L^2) the high word enters the hash too, so the
// mask does not repeat; below that it is zero and changes nothing
__device__ __forceinline__ unsigned att_hash(const unsigned long long idx, const unsigned s0, const unsigned s1) {
    unsigned x = (unsigned)idx ^ s0;
    x *= 0x9E3779B1u; x ^= x >> 15;
    x *= 0x85EBCA6Bu; x ^= x >> 13;
    x += s1 + (unsigned)(idx >> 32) * 0x9E3779B1u;
    x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
struct att_args {
    const float* qkv;
    const float* o;
    const float* dout;
    const float* lse_in;
    float* out;        // forward: o; backward: dqkv
    float* lse;        // forward: lse (N, H, L); backward (a): delta (N, H, L)
    const float* delta_in;
    int L, H;
    float scale, keep_inv;
    unsigned thr, s0, s1;
};

// rows [0, Lp) of (sequence n, head h) of one of q / k / v or of a (N, L, H, D) tensor -- `src` points at row 0, rows `row_stride` apart -- into a
// row-major LDS tile; rows >= L are zeros
template <int D, int NT>
__device__ __forceinline__ void att_fill(float* tile, const float* src, const size_t row_stride, const int L, const int Lp) {
    constexpr int KLD = D + 4;
    for (int i = threadIdx.x; i < Lp * (D / 4); i += NT) {
        const int row = i / (D / 4), d4 = i % (D / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < L) v = ald4(src + (size_t)row * row_stride + 4 * d4);
        *reinterpret_cast<float4*>(&tile[row * KLD + 4 * d4]) = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------------ forward
// One pass over the key blocks with a running maximum and sum per query (the scores of ONE block in registers: 16): the round-4 kernel held
// all blocks' scores at once -- 256 + 125 registers at 256 steps, one wave per SIMD, 174 us at SepFormer's intra-chunk shape
// (profiles/r07t_attention.txt).  A block whose maximum raises the running one rescales what has been accumulated: the accumulator's
// register r is query row(r, lk), whose factor lives in that LANE -- sixteen lane reads (ds_bpermute) per block, against sixteen
// MFMAs for a second pass over the scores.  NW waves of 32 queries share the K / V tiles of a (sequence, head).
template <int D, int ML, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const att_args p) {
    constexpr int DH = D / 2, KLD = D + 4;
    __shared__ __attribute__((aligned(16))) float Kr[ML * KLD + 32];
    __shared__ __attribute__((aligned(16))) float Vr[ML * KLD + 32];
    const int L = p.L, H = p.H, Lp = (L + 31) & ~31, nkb = Lp >> 5;
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 32 * NW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, lk = lane >> 5;
    const size_t RS = (size_t)3 * H * D;
    const float* base = p.qkv + (size_t)n * L * RS + (size_t)h * D;
    att_fill<D, 64 * NW>(Kr, base + (size_t)H * D, RS, L, Lp);
    att_fill<D, 64 * NW>(Vr, base + (size_t)2 * H * D, RS, L, Lp);
    __syncthreads();
    if (q0 + 32 * w >= L) return;                           // a wave without a live query (short sequences): it has helped to fill, no barrier follows
    const int q = q0 + 32 * w + l31;
    float qv[DH];
#pragma unroll
    for (int j = 0; j < DH / 4; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < L) v = ald4(base + (size_t)q * RS + lk * DH + 4 * j);
        const float sc = p.scale * ATT_LOG2E;
        qv[4 * j] = v.x * sc; qv[4 * j + 1] = v.y * sc; qv[4 * j + 2] = v.z * sc; qv[4 * j + 3] = v.w * sc;
    }
    const unsigned long long ebase = (((unsigned long long)n * H + h) * L + q) * (unsigned long long)L;
    float m = -INFINITY, sum = 0.f;                          // running maximum (common to lane and lane ^ 32) and this lane's part of the sum
    att_f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        att_f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = &Kr[(kb * 32 + l31) * KLD + lk * DH];
#pragma unroll
        for (int j = 0; j < DH / 4; ++j) {
            const float4 a = ald4(kp + 4 * j);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qv[4 * j], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qv[4 * j + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qv[4 * j + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qv[4 * j + 3], s, 0, 0, 0);
        }
        // this lane's registers hold the keys {kb * 32 + row(r, lk)}, lane ^ 32 the others
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = kb * 32 + att_row(r, lk) < L ? s[r] : -INFINITY;
            bm = fmaxf(bm, s[r]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);                       // finite: every block has a live key
        const float alpha = att_exp2(m - mn);                // (first block: 2^-inf = 0, and there is nothing to rescale)
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float e = att_exp2(s[r] - mn);                   // 2^-inf = 0 for the dead keys
            ps += e;
            if (p.thr != 0u) e = att_hash(ebase + (unsigned)(kb * 32 + att_row(r, lk)), p.s0, p.s1) >= p.thr ? e * p.keep_inv : 0.f;
            s[r] = e;
        }
        sum = fmaf(sum, alpha, ps);
        m = mn;
        if (kb > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[r] *= __shfl(alpha, att_row(r, lk), 64);
        }
        // O += P V: A = the (unnormalised) probabilities as they are (lane = query, k slot lk <-> key row(r, lk)), B = V[that key][dd = l31]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = Vr[(kb * 32 + att_row(r, lk)) * KLD + l31];              // (columns >= D: the next row's values -- their results are not stored)
            oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], b, oacc, 0, 0, 0);
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (lk == 0 && q < L) p.lse[((size_t)n * H + h) * L + q] = m * ATT_LN2 + logf(sum);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float f = __shfl(inv, att_row(r, lk), 64);
        const int qq = q0 + 32 * w + att_row(r, lk);
        if (l31 < D && qq < L) p.out[(((size_t)n * L + qq) * H + h) * D + l31] = oacc[r] * f;
    }
}

// ------------------------------------------------------------------------------------------------------------------------ backward (a): dQ, delta
template <int D, int ML, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_q_kernel(const att_args p) {
    constexpr int DH = D / 2, KLD = D + 4;
    __shared__ __attribute__((aligned(16))) float Kr[ML * KLD + 32];
    __shared__ __attribute__((aligned(16))) float Vr[ML * KLD + 32];
    const int L = p.L, H = p.H, Lp = (L + 31) & ~31, nkb = Lp >> 5;
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 32 * NW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, lk = lane >> 5;
    const size_t RS = (size_t)3 * H * D, OS = (size_t)H * D;
    const float* base = p.qkv + (size_t)n * L * RS + (size_t)h * D;
    att_fill<D, 64 * NW>(Kr, base + (size_t)H * D, RS, L, Lp);
    att_fill<D, 64 * NW>(Vr, base + (size_t)2 * H * D, RS, L, Lp);
    __syncthreads();
    if (q0 + 32 * w >= L) return;                           // (as in the forward kernel)
    const int q = q0 + 32 * w + l31;
    float qv[DH], dov[DH];
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < DH / 4; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), g = v, oo = v;
        if (q < L) {
            v = ald4(base + (size_t)q * RS + lk * DH + 4 * j);
            g = ald4(p.dout + ((size_t)n * L + q) * OS + (size_t)h * D + lk * DH + 4 * j);
            oo = ald4(p.o + ((size_t)n * L + q) * OS + (size_t)h * D + lk * DH + 4 * j);
        }
        const float sc = p.scale * ATT_LOG2E;
        qv[4 * j] = v.x * sc; qv[4 * j + 1] = v.y * sc; qv[4 * j + 2] = v.z * sc; qv[4 * j + 3] = v.w * sc;
        dov[4 * j] = g.x; dov[4 * j + 1] = g.y; dov[4 * j + 2] = g.z; dov[4 * j + 3] = g.w;
        dl += (g.x * oo.x + g.y * oo.y) + (g.z * oo.z + g.w * oo.w);
    }
    dl += __shfl_xor(dl, 32, 64);                           // delta_q = sum_dd dO[q][dd] O[q][dd]
    const size_t sidx = ((size_t)n * H + h) * L + q;
    float lse = 0.f;
    if (q < L) {
        lse = p.lse_in[sidx] * ATT_LOG2E;
        if (lk == 0) p.lse[sidx] = dl;
    }
    const unsigned long long ebase = (unsigned long long)sidx * (unsigned long long)L;
    att_f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        att_f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const float* kp = &Kr[(kb * 32 + l31) * KLD + lk * DH];
        const float* vp = &Vr[(kb * 32 + l31) * KLD + lk * DH];
#pragma unroll
        for (int j = 0; j < DH / 4; ++j) {
            const float4 a = ald4(kp + 4 * j), c = ald4(vp + 4 * j);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qv[4 * j], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, dov[4 * j], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qv[4 * j + 1], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, dov[4 * j + 1], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qv[4 * j + 2], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, dov[4 * j + 2], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qv[4 * j + 3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, dov[4 * j + 3], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + att_row(r, lk);
            const float pr = key < L ? att_exp2(s[r] - lse) : 0.f;
            float dpv = dp[r];
            if (p.thr != 0u) dpv = att_hash(ebase + (unsigned)key, p.s0, p.s1) >= p.thr ? dpv * p.keep_inv : 0.f;
            s[r] = pr * (dpv - dl);                          // dS
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = Kr[(kb * 32 + att_row(r, lk)) * KLD + l31];
            dq = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], b, dq, 0, 0, 0);
        }
    }
    if (l31 < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + 32 * w + att_row(r, lk);
            if (qq < L) p.out[((size_t)n * L + qq) * RS + (size_t)h * D + l31] = dq[r] * p.scale;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------ backward (b): dK, dV
template <int D, int ML, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kv_kernel(const att_args p) {
    constexpr int DH = D / 2, KLD = D + 4;
    __shared__ __attribute__((aligned(16))) float Qr[ML * KLD + 32];
    __shared__ __attribute__((aligned(16))) float Gr[ML * KLD + 32];      // dO
    __shared__ __attribute__((aligned(16))) float lse_s[ML];
    __shared__ __attribute__((aligned(16))) float del_s[ML];
    const int L = p.L, H = p.H, Lp = (L + 31) & ~31, nqb = Lp >> 5;
    const int n = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 32 * NW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, lk = lane >> 5;
    const size_t RS = (size_t)3 * H * D, OS = (size_t)H * D;
    const float* base = p.qkv + (size_t)n * L * RS + (size_t)h * D;
    att_fill<D, 64 * NW>(Qr, base, RS, L, Lp);
    att_fill<D, 64 * NW>(Gr, p.dout + (size_t)n * L * OS + (size_t)h * D, OS, L, Lp);
    for (int i = threadIdx.x; i < Lp; i += 64 * NW) {
        lse_s[i] = i < L ? p.lse_in[((size_t)n * H + h) * L + i] * ATT_LOG2E : 0.f;
        del_s[i] = i < L ? p.delta_in[((size_t)n * H + h) * L + i] : 0.f;
    }
    __syncthreads();
    if (k0 + 32 * w >= L) return;                           // a wave without a live key
    const int key = k0 + 32 * w + l31;
    float kv[DH], vv[DH];
#pragma unroll
    for (int j = 0; j < DH / 4; ++j) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        if (key < L) {
            a = ald4(base + (size_t)key * RS + (size_t)H * D + lk * DH + 4 * j);
            c = ald4(base + (size_t)key * RS + (size_t)2 * H * D + lk * DH + 4 * j);
        }
        const float sc = p.scale * ATT_LOG2E;
        kv[4 * j] = a.x * sc; kv[4 * j + 1] = a.y * sc; kv[4 * j + 2] = a.z * sc; kv[4 * j + 3] = a.w * sc;
        vv[4 * j] = c.x; vv[4 * j + 1] = c.y; vv[4 * j + 2] = c.z; vv[4 * j + 3] = c.w;
    }
    const size_t hbase = ((size_t)n * H + h) * L;
    att_f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
    for (int qb = 0; qb < nqb; ++qb) {
        att_f32x16 s, dp;                                    // rows = queries (registers), columns = keys (lanes)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const float* qp = &Qr[(qb * 32 + l31) * KLD + lk * DH];
        const float* gp = &Gr[(qb * 32 + l31) * KLD + lk * DH];
#pragma unroll
        for (int j = 0; j < DH / 4; ++j) {
            const float4 a = ald4(qp + 4 * j), c = ald4(gp + 4 * j);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kv[4 * j], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, vv[4 * j], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kv[4 * j + 1], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, vv[4 * j + 1], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kv[4 * j + 2], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, vv[4 * j + 2], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kv[4 * j + 3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, vv[4 * j + 3], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = qb * 32 + att_row(r, lk);
            const bool live = qq < L && key < L;
            const float pr = live ? att_exp2(s[r] - lse_s[qq]) : 0.f;
            float pd = pr, dpv = dp[r];
            if (p.thr != 0u) {
                const bool keep = att_hash((unsigned long long)(hbase + qq) * (unsigned long long)L + (unsigned)key, p.s0, p.s1) >= p.thr;
                pd = keep ? pr * p.keep_inv : 0.f;
                dpv = keep ? dpv * p.keep_inv : 0.f;
            }
            dp[r] = pd;                                      // Pd
            s[r] = pr * (dpv - del_s[qq]);                   // dS
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = qb * 32 + att_row(r, lk);
            const float bg = Gr[qq * KLD + l31], bq = Qr[qq * KLD + l31];
            dv = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[r], bg, dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], bq, dk, 0, 0, 0);
        }
    }
    if (l31 < D) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + 32 * w + att_row(r, lk);
            if (kk < L) {
                float* dst = p.out + ((size_t)n * L + kk) * RS + (size_t)h * D + l31;
                dst[(size_t)H * D] = dk[r] * p.scale;
                dst[(size_t)2 * H * D] = dv[r];
            }
        }
    }
}

bool att_shape_ok(int N, int L, int H, int D) {
    return N > 0 && N <= 65535 && H > 0 && H <= 65535 && L > 0 && L <= ATT_MAXL && (D == 8 || D == 16 || D == 32);
}
att_args att_make(const float* qkv, const float* o, const float* dout, const float* lse_in, const float* delta_in, float* out, float* lse, int L, int H,
                  float scale, float p_drop, unsigned long long seed) {
    att_args a;
    a.qkv = qkv; a.o = o; a.dout = dout; a.lse_in = lse_in; a.delta_in = delta_in; a.out = out; a.lse = lse;
    a.L = L; a.H = H; a.scale = scale;
    double t = (double)p_drop * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    a.thr = p_drop > 0.f ? (unsigned)t : 0u;
    if (p_drop > 0.f && a.thr == 0u) a.thr = 1u;
    a.keep_inv = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    a.s0 = (unsigned)(seed & 0xffffffffull);
    a.s1 = (unsigned)(seed >> 32);
    return a;
}

}  // namespace

/* q (N, L, 3, H, D) packed -> o (N, L, H, D), lse (N, H, L) */
extern "C" int sep_attn_fwd(const float* qkv, float* o, float* lse, int N, int L, int H, int D, float scale, float p_drop, unsigned long long seed,
                            sep_stream_t stream) {
    SEP_REQUIRE(qkv && o && lse, "sep_attn_fwd: null pointer");
    SEP_REQUIRE(att_shape_ok(N, L, H, D), "sep_attn_fwd: N=%d L=%d H=%d D=%d (L <= 320, D in {8, 16, 32})", N, L, H, D);
    SEP_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sep_attn_fwd: dropout rate %g", (double)p_drop);
    const att_args a = att_make(qkv, nullptr, nullptr, nullptr, nullptr, o, lse, L, H, scale, p_drop, seed);
    /* waves per workgroup = 32-query blocks that share one K / V fill: all of a sequence up to 256 steps */
#define SEP_ATF_ONE(DD, MLL, NWW) hipLaunchKernelGGL((attn_fwd_kernel<DD, MLL, NWW>), dim3((L + 32 * NWW - 1) / (32 * NWW), H, N), dim3(64 * NWW), 0, (hipStream_t)stream, a)
#define SEP_ATF(DD)                                  \
    do {                                             \
        if (L <= 64) SEP_ATF_ONE(DD, 64, 2);         \
        else if (L <= 128) SEP_ATF_ONE(DD, 128, 4);  \
        else if (L <= 256) SEP_ATF_ONE(DD, 256, 8);  \
        else SEP_ATF_ONE(DD, 320, 8);                \
    } while (0)
    if (D == 8) SEP_ATF(8); else if (D == 16) SEP_ATF(16); else SEP_ATF(32);
#undef SEP_ATF_ONE
#undef SEP_ATF
    SEP_CHECK_LAUNCH("sep_attn_fwd");
    return 0;
}

/* dqkv (N, L, 3, H, D) is written completely; delta (N, H, L) is scratch */
extern "C" int sep_attn_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* delta, float* dqkv, int N, int L, int H, int D,
                            float scale, float p_drop, unsigned long long seed, sep_stream_t stream) {
    SEP_REQUIRE(qkv && o && dout && lse && delta && dqkv, "sep_attn_bwd: null pointer");
    SEP_REQUIRE(att_shape_ok(N, L, H, D), "sep_attn_bwd: N=%d L=%d H=%d D=%d (L <= 320, D in {8, 16, 32})", N, L, H, D);
    SEP_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "sep_attn_bwd: dropout rate %g", (double)p_drop);
    const att_args a = att_make(qkv, o, dout, lse, delta, dqkv, delta, L, H, scale, p_drop, seed);
#define SEP_ATB_ONE(DD, MLL, NWW)                                                                                                                             \
    do {                                                                                                                                                      \
        const dim3 grid((L + 32 * NWW - 1) / (32 * NWW), H, N);                                                                                               \
        hipLaunchKernelGGL((attn_bwd_q_kernel<DD, MLL, NWW>), grid, dim3(64 * NWW), 0, (hipStream_t)stream, a);                                               \
        hipLaunchKernelGGL((attn_bwd_kv_kernel<DD, MLL, NWW>), grid, dim3(64 * NWW), 0, (hipStream_t)stream, a);                                              \
    } while (0)
#define SEP_ATB(DD)                                  \
    do {                                             \
        if (L <= 64) SEP_ATB_ONE(DD, 64, 2);         \
        else if (L <= 128) SEP_ATB_ONE(DD, 128, 4);  \
        else if (L <= 256) SEP_ATB_ONE(DD, 256, 8);  \
        else SEP_ATB_ONE(DD, 320, 8);                \
    } while (0)
    if (D == 8) SEP_ATB(8); else if (D == 16) SEP_ATB(16); else SEP_ATB(32);
#undef SEP_ATB_ONE
#undef SEP_ATB
    SEP_CHECK_LAUNCH("sep_attn_bwd");
    return 0;
}
