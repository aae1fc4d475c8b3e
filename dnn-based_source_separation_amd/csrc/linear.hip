// Token-major dense layers of the dual-path separators in fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32): the input projection of
// the LSTMs, the Linear layer behind them, and their input / weight gradients.  Activations are [tokens][features] (features contiguous,
// tokens = sequences x steps: 128,500 per utterance pair at the DPRNN-TasNet recipe's sizes), weights are torch's [out][in]:
//
//   sep_linear_fwd          y[t][n]  = sum_k x[t][k] w[n][k] + bias[n] (+ bias2[n])                  torch.addmm(b, x, w.t())
//   sep_linear_bwd_input    dx[t][k] (+)= sum_n dy[t][n] w[n][k]                                       dy @ w
//   sep_linear_bwd_weight   partial[s][n][k] = sum over the tokens of slab s of dy[t][n] x[t + shift][k]   dy.t() @ x, in `nslab` partial sums (x rows ldx apart)
//                           partial_bias[s][n] = sum over the same tokens of dy[t][n]                     dy.sum(0)
//     (shift = -1 / +1 with sequences of L steps: x is the PREVIOUS / NEXT step's row of the same sequence, zero at the sequence's first /
//      last step -- the h_{t-1} operand of the recurrent weights' gradient without materialising it)
//
// Replaces the library GEMMs behind reference src/models/dprnn.py:65-148 (nn.LSTM's input projections, the nn.Linear of IntraChunkRNN /
// InterChunkRNN) -- hipBLASLt's picks for these tall, skinny fp32 shapes ran at 8 TFLOP/s and a fifth of the HBM rate (rocprofv3 of the
// DPRNN-TasNet step: 30 of 56 ms; profiles/r03s_dprnn_kernel_stats.md).  One kernel: C[i][j] = sum_r A(i, r) B(r, j) on a TI x TJ
// workgroup tile, 32 contraction steps at a time through LDS tiles that are contiguous along r for BOTH operands ([row][32 + 4]:
// ds_read_b128 operand fetches, conflict-free at a row pitch of 144 bytes), the three products differing only in how a tile is gathered
// from global memory (straight copy, or transposed on the way in).
#include "common.hpp"

namespace {

typedef float lin_f32x16 __attribute__((ext_vector_type(16)));
constexpr int LIN_RC = 32;          // contraction steps per LDS tile
constexpr int LIN_LD = LIN_RC + 4;  // row pitch in floats
enum { LIN_FWD = 0, LIN_BWD_INPUT = 1, LIN_BWD_WEIGHT = 2 };

struct lin_args {
    const float* a;        // FWD: x; BWD_INPUT: dy; BWD_WEIGHT: dy
    const float* b;        // FWD: w; BWD_INPUT: w;  BWD_WEIGHT: x
    const float* bias;
    const float* bias2;
    float* c;              // y / dx / partial
    float* cbias;          // BWD_WEIGHT: partial_bias (may be null)
    long ntok;
    int K, N;              // features in / out of the layer
    int L, shift;          // BWD_WEIGHT: sequence length and row shift of x
    int nslab;
    int accumulate;
    long ldb;              // BWD_WEIGHT: row stride of x
};

template <int MODE, int TI, int TJ>
__global__ __launch_bounds__(256) void linear_kernel(const lin_args p) {
    constexpr int RB = TI / 32;                // 32-row blocks of the tile
    constexpr int CW = 4 / RB;                 // waves side by side along j
    constexpr int NB = TJ / 32 / CW;           // 32-column blocks per wave
    static_assert(RB == 2 || RB == 4, "tile rows");
    static_assert(NB >= 1 && NB * CW * 32 == TJ, "tile columns");
    __shared__ __attribute__((aligned(16))) float As[TI * LIN_LD];
    __shared__ __attribute__((aligned(16))) float Bs[TJ * LIN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid % RB, wc = wid / RB;
    const int l31 = lane & 31, lk = lane >> 5;

    // tile coordinates and the contraction range
    long i0, r_begin, r_end;
    int j0, slab = 0;
    if (MODE == LIN_BWD_WEIGHT) {
        const int ntj = p.K / TJ, nti = p.N / TI;
        const int tile = blockIdx.x % (ntj * nti);
        slab = blockIdx.x / (ntj * nti);
        i0 = (long)(tile / ntj) * TI;
        j0 = (tile % ntj) * TJ;
        const long chunks = (p.ntok + LIN_RC - 1) / LIN_RC;
        const long per = (chunks + p.nslab - 1) / p.nslab;
        r_begin = (long)slab * per * LIN_RC;
        r_end = r_begin + per * LIN_RC;
        if (r_end > p.ntok) r_end = p.ntok;
    } else {
        const int nout = MODE == LIN_FWD ? p.N : p.K;
        const int ntj = nout / TJ;
        i0 = (long)(blockIdx.x / ntj) * TI;
        j0 = (blockIdx.x % ntj) * TJ;
        r_begin = 0;
        r_end = MODE == LIN_FWD ? p.K : p.N;
    }

    lin_f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float bsum = 0.f;                                     // BWD_WEIGHT: this thread's share of a row sum of dy (thread = (row, half))

    // A tile [TI][32] and B tile [TJ][32], both contiguous along the contraction: a straight copy of 128-byte row pieces where the operand
    // is stored that way, float4s along the OTHER index scattered into four rows where it is not.  Granule g (4 steps) of row `row` sits
    // at granule g ^ ((row >> 4) & 3): the scattered writes of 32 lanes then fall on 16 banks instead of 4 (rows 4 apart are 16 banks
    // apart at this pitch), and the operand reads -- 16 consecutive rows per quarter wave, one swizzle value -- stay conflict-free.
    // The global loads of chunk c + 1 are issued before the MFMAs of chunk c and land in registers (fa / fb) behind them.
    constexpr int QA = TI / 32, QB = TJ / 32;
    float4 fa[QA], fb[QB];
    auto fetch = [&](const long r0) {
        if (MODE == LIN_FWD || MODE == LIN_BWD_INPUT) {
            const int lda = MODE == LIN_FWD ? p.K : p.N;          // A(i, r) = a[i0 + i][r0 + r]
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const long gi = i0 + (tid >> 3) + 32 * q;
                fa[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gi < p.ntok) fa[q] = *reinterpret_cast<const float4*>(p.a + gi * lda + r0 + 4 * (tid & 7));
            }
        } else {                                                  // A(i, r) = dy[r0 + r][i0 + i]: float4 along i
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const long gt = r0 + (tid / (TI / 4)) + (256 / (TI / 4)) * q;
                fa[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gt < r_end) fa[q] = *reinterpret_cast<const float4*>(p.a + gt * p.N + i0 + 4 * (tid % (TI / 4)));
            }
        }
        if (MODE == LIN_FWD) {                                    // B(r, j) = w[j0 + j][r0 + r]
#pragma unroll
            for (int q = 0; q < QB; ++q)
                fb[q] = *reinterpret_cast<const float4*>(p.b + (size_t)(j0 + (tid >> 3) + 32 * q) * p.K + r0 + 4 * (tid & 7));
        } else {                                                  // B(r, j) = w[r0 + r][j0 + j] or the (shifted) x[r0 + r][j0 + j]: float4 along j
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int rr = (tid / (TJ / 4)) + (256 / (TJ / 4)) * q, jq = tid % (TJ / 4);
                fb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == LIN_BWD_INPUT) {
                    fb[q] = *reinterpret_cast<const float4*>(p.b + (size_t)(r0 + rr) * p.K + j0 + 4 * jq);
                } else {
                    const long gt = r0 + rr;
                    bool ok = gt < r_end;
                    long src = gt;
                    if (p.shift != 0) {
                        const int step = (int)(gt % p.L) + p.shift;
                        ok = ok && step >= 0 && step < p.L;
                        src = gt + p.shift;
                    }
                    if (ok) fb[q] = *reinterpret_cast<const float4*>(p.b + src * p.ldb + j0 + 4 * jq);
                }
            }
        }
    };
    auto put_row = [&](float* tile, const int row, const int c4, const float4 v) {           // 4 steps of one row
        *reinterpret_cast<float4*>(&tile[row * LIN_LD + 4 * (c4 ^ ((row >> 4) & 3))]) = v;
    };
    auto put_col = [&](float* tile, const int row4, const int rr, const float4 v) {          // step rr of rows row4 .. row4 + 3
        const int c = 4 * ((rr >> 2) ^ ((row4 >> 4) & 3)) + (rr & 3);
        tile[(row4 + 0) * LIN_LD + c] = v.x; tile[(row4 + 1) * LIN_LD + c] = v.y;
        tile[(row4 + 2) * LIN_LD + c] = v.z; tile[(row4 + 3) * LIN_LD + c] = v.w;
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            if (MODE == LIN_FWD || MODE == LIN_BWD_INPUT) put_row(As, (tid >> 3) + 32 * q, tid & 7, fa[q]);
            else put_col(As, 4 * (tid % (TI / 4)), (tid / (TI / 4)) + (256 / (TI / 4)) * q, fa[q]);
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if (MODE == LIN_FWD) put_row(Bs, (tid >> 3) + 32 * q, tid & 7, fb[q]);
            else put_col(Bs, 4 * (tid % (TJ / 4)), (tid / (TJ / 4)) + (256 / (TJ / 4)) * q, fb[q]);
        }
    };

    if (r_begin < r_end) fetch(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += LIN_RC) {
        commit();
        __syncthreads();
        if (r0 + LIN_RC < r_end) fetch(r0 + LIN_RC);
        if (MODE == LIN_BWD_WEIGHT && p.cbias != nullptr && j0 == 0 && tid < 2 * TI) {      // row sums of dy: thread = (row, 16-step half; the swizzle stays inside a half)
            const float* rp = &As[(tid >> 1) * LIN_LD + 16 * (tid & 1)];
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += rp[e];
            bsum += s;
        }
        // ---- 16 k-steps of 2: a lane's float4 holds the contraction steps {4g + 16 lk .. + 3} of its row -- the pairing of steps into
        // the instruction's two k slots is free as long as A and B agree on it
        const int swa = ((32 * wr + l31) >> 4) & 3;              // the swizzle of this lane's rows
        const float* ap = &As[(32 * wr + l31) * LIN_LD + 16 * lk];
        const float* bp = &Bs[(32 * NB * wc + l31) * LIN_LD + 16 * lk];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a4 = *reinterpret_cast<const float4*>(ap + 4 * (g ^ swa));
            float4 b4[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int swb = ((32 * (NB * wc + n) + l31) >> 4) & 3;
                b4[n] = *reinterpret_cast<const float4*>(bp + 32 * n * LIN_LD + 4 * (g ^ swb));
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[n].x, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[n].y, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[n].z, acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[n].w, acc[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- results: register r of a lane is row (r % 4) + 8 (r / 4) + 4 lk, column l31 of its 32 x 32 block
    const int ldc = MODE == LIN_FWD ? p.N : p.K;
    float* cbase = p.c + (MODE == LIN_BWD_WEIGHT ? (size_t)slab * p.N * p.K : (size_t)0);
    const long nrows = MODE == LIN_BWD_WEIGHT ? (long)p.N : p.ntok;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int col = j0 + 32 * (NB * wc + n) + l31;
        float add = 0.f;
        if (MODE == LIN_FWD) {
            if (p.bias) add += p.bias[col];
            if (p.bias2) add += p.bias2[col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = i0 + 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < nrows) {
                float* dst = cbase + row * ldc + col;
                float v = acc[n][r] + add;
                if (MODE == LIN_BWD_INPUT && p.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
    if (MODE == LIN_BWD_WEIGHT && p.cbias != nullptr && j0 == 0 && tid < 2 * TI) {
        const float tot = bsum + __shfl_xor(bsum, 1, 64);
        if ((tid & 1) == 0) p.cbias[(size_t)slab * p.N + i0 + (tid >> 1)] = tot;
    }
}


// (B, F, S, K) chunked features <-> token-major rows [B][tokens][F] for the dual-path recurrences (reference src/models/dprnn.py:73-76,
// 123-126: `input.permute(0, 2, 3, 1).reshape(B*S, K, F)` for the intra-chunk path, `input.permute(0, 3, 2, 1).reshape(B*K, S, F)` for the
// inter-chunk path, and their inverses behind the Linear): 32 x 32 tiles through LDS, 128-byte rows on both sides -- torch's strided
// copies ran at ~1 TB/s here (61 - 93 us for 2 x 33 MB, 66 launches per DPRNN-TasNet step).
template <bool TO_TOKENS>
__global__ __launch_bounds__(256) void chunk_tokens_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int S, int K, int inter) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int k0 = blockIdx.x * 32, s = blockIdx.y;
    const int nft = (F + 31) / 32;
    const int b = blockIdx.z / nft, f0 = (blockIdx.z % nft) * 32;
    const size_t tokbase = (size_t)b * S * K;
    auto tok = [&](int k) { return inter ? (size_t)k * S + s : (size_t)s * K + k; };
    if (TO_TOKENS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = f0 + ty + 8 * r, k = k0 + tx;
            tile[ty + 8 * r][tx] = (f < F && k < K) ? src[(((size_t)b * F + f) * S + s) * K + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + ty + 8 * r, f = f0 + tx;
            if (k < K && f < F) dst[(tokbase + tok(k)) * F + f] = tile[tx][ty + 8 * r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + ty + 8 * r, f = f0 + tx;
            tile[ty + 8 * r][tx] = (k < K && f < F) ? src[(tokbase + tok(k)) * F + f] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = f0 + ty + 8 * r, k = k0 + tx;
            if (f < F && k < K) dst[(((size_t)b * F + f) * S + s) * K + k] = tile[tx][ty + 8 * r];
        }
    }
}

template <int MODE>
int lin_launch(const lin_args& p, int ti, int tj, unsigned grid, hipStream_t stream) {
#define SEP_LIN(TI, TJ) hipLaunchKernelGGL((linear_kernel<MODE, TI, TJ>), dim3(grid), dim3(256), 0, stream, p)
    if (ti == 128 && tj == 128) SEP_LIN(128, 128);
    else if (ti == 128 && tj == 64) SEP_LIN(128, 64);
    else if (ti == 64 && tj == 128) SEP_LIN(64, 128);
    else SEP_LIN(64, 64);
#undef SEP_LIN
    return 0;
}

}  // namespace

extern "C" int sep_linear_fwd(const float* x, const float* w, const float* bias, const float* bias2, float* y, long ntok, int K, int N,
                              sep_stream_t stream) {
    SEP_REQUIRE(x && w && y, "sep_linear_fwd: null pointer");
    SEP_REQUIRE(ntok > 0 && K > 0 && N > 0 && K % 32 == 0 && N % 64 == 0, "sep_linear_fwd: K=%d must be a multiple of 32 and N=%d of 64", K, N);
    lin_args p = {x, w, bias, bias2, y, nullptr, ntok, K, N, 1, 0, 1, 0, K};
    const int tj = N % 128 == 0 ? 128 : 64;
    const long tiles = ((ntok + 127) / 128) * (N / tj);
    SEP_REQUIRE(tiles <= 0x7fffffffL, "sep_linear_fwd: too many tiles");
    lin_launch<LIN_FWD>(p, 128, tj, (unsigned)tiles, (hipStream_t)stream);
    SEP_CHECK_LAUNCH("sep_linear_fwd");
    return 0;
}

extern "C" int sep_linear_bwd_input(const float* dy, const float* w, float* dx, long ntok, int K, int N, int accumulate, sep_stream_t stream) {
    SEP_REQUIRE(dy && w && dx, "sep_linear_bwd_input: null pointer");
    SEP_REQUIRE(ntok > 0 && K > 0 && N > 0 && N % 32 == 0 && K % 64 == 0, "sep_linear_bwd_input: N=%d must be a multiple of 32 and K=%d of 64", N, K);
    lin_args p = {dy, w, nullptr, nullptr, dx, nullptr, ntok, K, N, 1, 0, 1, accumulate, K};
    const int tj = K % 128 == 0 ? 128 : 64;
    const long tiles = ((ntok + 127) / 128) * (K / tj);
    SEP_REQUIRE(tiles <= 0x7fffffffL, "sep_linear_bwd_input: too many tiles");
    lin_launch<LIN_BWD_INPUT>(p, 128, tj, (unsigned)tiles, (hipStream_t)stream);
    SEP_CHECK_LAUNCH("sep_linear_bwd_input");
    return 0;
}

extern "C" int sep_linear_bwd_weight(const float* dy, const float* x, long ldx, float* partial, float* partial_bias, long ntok, int K, int N, int L,
                                     int shift, int nslab, sep_stream_t stream) {
    SEP_REQUIRE(dy && x && partial, "sep_linear_bwd_weight: null pointer");
    SEP_REQUIRE(ntok > 0 && K > 0 && N > 0 && N % 64 == 0 && K % 64 == 0, "sep_linear_bwd_weight: N=%d and K=%d must be multiples of 64", N, K);
    SEP_REQUIRE(nslab >= 1 && shift >= -1 && shift <= 1 && L >= 1 && (shift == 0 || ntok % L == 0),
                "sep_linear_bwd_weight: nslab >= 1, shift in {-1, 0, 1}, and whole sequences of L steps for a shifted x");
    SEP_REQUIRE(ldx >= K && ldx % 4 == 0, "sep_linear_bwd_weight: ldx=%ld must be a multiple of 4 and >= K=%d", ldx, K);
    lin_args p = {dy, x, nullptr, nullptr, partial, partial_bias, ntok, K, N, L, shift, nslab, 0, ldx};
    const int ti = N % 128 == 0 ? 128 : 64, tj = K % 128 == 0 ? 128 : 64;
    const long grid = (long)(N / ti) * (K / tj) * nslab;
    SEP_REQUIRE(grid <= 0x7fffffffL, "sep_linear_bwd_weight: too many workgroups");
    lin_launch<LIN_BWD_WEIGHT>(p, ti, tj, (unsigned)grid, (hipStream_t)stream);
    SEP_CHECK_LAUNCH("sep_linear_bwd_weight");
    return 0;
}

static int chunk_tokens(const float* src, float* dst, int B, int F, int S, int K, int inter, bool to_tokens, const char* name, hipStream_t stream) {
    SEP_REQUIRE(src && dst && B > 0 && F > 0 && S > 0 && K > 0, "%s: bad arguments", name);
    const long gz = (long)B * ((F + 31) / 32);
    SEP_REQUIRE(S <= 65535 && gz <= 65535, "%s: S = %d chunks and B x ceil(F / 32) = %ld must fit a grid dimension (65535)", name, S, gz);
    const dim3 grid((K + 31) / 32, S, (unsigned)gz);
    if (to_tokens) hipLaunchKernelGGL((chunk_tokens_kernel<true>), grid, dim3(256), 0, stream, src, dst, F, S, K, inter);
    else hipLaunchKernelGGL((chunk_tokens_kernel<false>), grid, dim3(256), 0, stream, src, dst, F, S, K, inter);
    return 0;
}

extern "C" int sep_chunk_to_tokens(const float* x, float* y, int B, int F, int S, int K, int inter, sep_stream_t stream) {
    if (chunk_tokens(x, y, B, F, S, K, inter, true, "sep_chunk_to_tokens", (hipStream_t)stream)) return -1;
    SEP_CHECK_LAUNCH("sep_chunk_to_tokens");
    return 0;
}

extern "C" int sep_tokens_to_chunk(const float* y, float* x, int B, int F, int S, int K, int inter, sep_stream_t stream) {
    if (chunk_tokens(y, x, B, F, S, K, inter, false, "sep_tokens_to_chunk", (hipStream_t)stream)) return -1;
    SEP_CHECK_LAUNCH("sep_tokens_to_chunk");
    return 0;
}
