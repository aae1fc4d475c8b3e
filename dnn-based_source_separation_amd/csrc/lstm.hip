// Recurrent sweeps of one LSTM direction for the dual-path RNN (reference src/models/dprnn.py:9-148: nn.LSTM inside
// IntraChunkRNN / InterChunkRNN).  The input projection x W_ih^T + b and the three weight-gradient products are plain
// library GEMMs on the host side; what cannot be a library call is the time recurrence, and that is what lives here:
//
//   forward :  a_t = xg_t + W_hh h_{t-1};  i,f,o = sigmoid, g = tanh;  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t)
//   backward:  reverse sweep carrying (dh, dc); emits the pre-activation gate gradients da_t (= d xg_t)
//
// One workgroup = 16 sequences x the whole hidden state, persistent over all L steps.  W_hh (4H x H fp32 = 256 KiB at
// H = 128) does not fit LDS; it lives in REGISTERS: H/16 waves, wave w owns hidden units [16w, 16w+16) and holds the
// 64 x H slice [i;f;g;o rows of its units] as MFMA A-fragments (4 x H/4 VGPRs).  Per step a wave runs 4 x H/4
// v_mfma_f32_16x16x4_f32 against h_{t-1} (B operand, 8 KiB in LDS, double-buffered), and the C layout leaves all four
// gates of (unit, sequence) in ONE lane, so the cell update is register-local; only h_t crosses waves (one barrier per
// step).  The backward sweep is the mirror image with W_hh^T slices and the 4H x 16 gate-gradient panel in LDS.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_f(float x) {
    // tanh(x) = 1 - 2 / (1 + e^{2x}); exp2 overflow -> inf -> rcp 0 -> 1, underflow -> 0 -> -1
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int LSTM_NS = 16;     // sequences per workgroup = N of the 16x16x4 MFMA

template <int H>
__global__ __launch_bounds__(H * 4) void lstm_fwd_kernel(const float* __restrict__ xg, const float* __restrict__ whh,
                                                         float* __restrict__ hout, float* __restrict__ gates,
                                                         float* __restrict__ cstate, int nseq, int L, int dirs, int hrow) {
    constexpr int KS = H / 4;                  // MFMA k-steps per gate block
    // dirs: 0 forward in time, 1 backward in time, 2 both -- blockIdx.y picks the direction and its slab of every buffer
    const int dir = dirs == 2 ? (int)blockIdx.y : 0;
    const int reverse = dirs == 2 ? dir : dirs;
    {
        const size_t rows = (size_t)nseq * L;
        xg += dir * rows * 4 * H; whh += (size_t)dir * 4 * H * H; hout += hrow == H ? dir * rows * H : dir * H;
        if (gates) gates += dir * rows * 4 * H;
        if (cstate) cstate += dir * rows * H;
    }
    __shared__ float hs[2][H * LSTM_NS];       // h_{t-1} / h_t as [unit][sequence]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;    // MFMA column (sequence) / row group
    const int seq0 = blockIdx.x * LSTM_NS;
    const int seq = seq0 + j;
    const bool live = seq < nseq;
    const int seqc = live ? seq : nseq - 1;
    const int u0 = 16 * w + 4 * q;             // this lane owns hidden units u0 .. u0+3 of sequence `seq`

    // A fragments: wf[g][ks] = W_hh[g*H + 16w + (lane & 15)][4 ks + (lane >> 4)]
    float wf[4][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[g][ks] = whh[(size_t)(g * H + 16 * w + j) * H + 4 * ks + q];
    for (int i = threadIdx.x; i < H * LSTM_NS; i += H * 4) hs[0][i] = 0.f;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const size_t seq_base = (size_t)seqc * L;
    auto load_x = [&](int t, float4 (&dst)[4]) {
        const float* p = xg + (seq_base + t) * 4 * H + u0;
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = ld4g(p + g * H);
    };
    float4 xc[4], xn[4];
    load_x(reverse ? L - 1 : 0, xc);
    for (int step = 0; step < L; ++step) {
        const int t = reverse ? L - 1 - step : step;
        const int cur = step & 1;
        if (step + 1 < L) load_x(reverse ? t - 1 : t + 1, xn);          // next step's projection, in flight under the MFMAs
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* hp = hs[cur];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float b = hp[(4 * ks + q) * LSTM_NS + j];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][ks], b, acc[g], 0, 0, 0);
        }
        const float xi[4] = {xc[0].x, xc[0].y, xc[0].z, xc[0].w}, xf[4] = {xc[1].x, xc[1].y, xc[1].z, xc[1].w};
        const float xgg[4] = {xc[2].x, xc[2].y, xc[2].z, xc[2].w}, xo[4] = {xc[3].x, xc[3].y, xc[3].z, xc[3].w};
        float gi[4], gf[4], gg[4], go[4], hn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gi[r] = sigmoid_f(acc[0][r] + xi[r]);
            gf[r] = sigmoid_f(acc[1][r] + xf[r]);
            gg[r] = tanh_f(acc[2][r] + xgg[r]);
            go[r] = sigmoid_f(acc[3][r] + xo[r]);
            c[r] = fmaf(gf[r], c[r], gi[r] * gg[r]);
            hn[r] = go[r] * tanh_f(c[r]);
            hs[cur ^ 1][(u0 + r) * LSTM_NS + j] = hn[r];
        }
        if (live) {
            const size_t o = seq_base + t;
            st4g(hout + o * hrow + u0, make_float4(hn[0], hn[1], hn[2], hn[3]));
            if (gates) {
                float* gp = gates + o * 4 * H + u0;
                st4g(gp, make_float4(gi[0], gi[1], gi[2], gi[3]));
                st4g(gp + H, make_float4(gf[0], gf[1], gf[2], gf[3]));
                st4g(gp + 2 * H, make_float4(gg[0], gg[1], gg[2], gg[3]));
                st4g(gp + 3 * H, make_float4(go[0], go[1], go[2], go[3]));
            }
            if (cstate) st4g(cstate + o * H + u0, make_float4(c[0], c[1], c[2], c[3]));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) xc[g] = xn[g];
        __syncthreads();                       // h_t complete in hs[cur^1]; everyone is done reading hs[cur]
    }
}

// Reverse sweep.  dh_out[t] is the gradient arriving at h_t from above; (dhr, dc) is what flows back from step t+1.
template <int H>
__global__ __launch_bounds__(H * 4) void lstm_bwd_kernel(const float* __restrict__ dhout, const float* __restrict__ gates,
                                                         const float* __restrict__ cstate, const float* __restrict__ whh,
                                                         float* __restrict__ dxg, int nseq, int L, int dirs, int hrow) {
    constexpr int KS = 4 * H / 4;              // k-steps over the 4H gate rows
    const int dir = dirs == 2 ? (int)blockIdx.y : 0;
    const int reverse = dirs == 2 ? dir : dirs;
    {
        const size_t rows = (size_t)nseq * L;
        dhout += hrow == H ? dir * rows * H : dir * H; gates += dir * rows * 4 * H; cstate += dir * rows * H; whh += (size_t)dir * 4 * H * H;
        dxg += dir * rows * 4 * H;
    }
    __shared__ float das[4 * H * LSTM_NS];     // d(pre-activation) of the current step as [gate row][sequence]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int seq = blockIdx.x * LSTM_NS + j;
    const bool live = seq < nseq;
    const int seqc = live ? seq : nseq - 1;
    const int u0 = 16 * w + 4 * q;

    // A fragments of W_hh^T: wt[ks] = W_hh[4 ks + (lane >> 4)][16 w + (lane & 15)]   (rows = this wave's hidden units)
    float wt[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wt[ks] = whh[(size_t)(4 * ks + q) * H + 16 * w + j];
    float dhr[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t seq_base = (size_t)seqc * L;

    for (int step = 0; step < L; ++step) {
        const int t = reverse ? step : L - 1 - step;               // the forward sweep's LAST step first
        const int tprev = reverse ? t + 1 : t - 1;                 // the step that fed c_{t-1}
        const bool has_prev = tprev >= 0 && tprev < L;
        const size_t o = seq_base + t;
        const float4 dho = ld4g(dhout + o * hrow + u0);
        const float* gp = gates + o * 4 * H + u0;
        const float4 vi = ld4g(gp), vf = ld4g(gp + H), vg = ld4g(gp + 2 * H), vo = ld4g(gp + 3 * H);
        const float4 vc = ld4g(cstate + o * H + u0);
        const float4 vcp = has_prev ? ld4g(cstate + (seq_base + tprev) * H + u0) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float dh4[4] = {dho.x, dho.y, dho.z, dho.w};
        const float gi[4] = {vi.x, vi.y, vi.z, vi.w}, gf[4] = {vf.x, vf.y, vf.z, vf.w};
        const float gg[4] = {vg.x, vg.y, vg.z, vg.w}, go[4] = {vo.x, vo.y, vo.z, vo.w};
        const float cc[4] = {vc.x, vc.y, vc.z, vc.w}, cp[4] = {vcp.x, vcp.y, vcp.z, vcp.w};
        float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = dh4[r] + dhr[r];
            const float tc = tanh_f(cc[r]);
            const float dcc = fmaf(dh * go[r], 1.f - tc * tc, dc[r]);
            dao[r] = dh * tc * go[r] * (1.f - go[r]);
            dai[r] = dcc * gg[r] * gi[r] * (1.f - gi[r]);
            daf[r] = dcc * cp[r] * gf[r] * (1.f - gf[r]);
            dag[r] = dcc * gi[r] * (1.f - gg[r] * gg[r]);
            dc[r] = dcc * gf[r];
            das[(0 * H + u0 + r) * LSTM_NS + j] = dai[r];
            das[(1 * H + u0 + r) * LSTM_NS + j] = daf[r];
            das[(2 * H + u0 + r) * LSTM_NS + j] = dag[r];
            das[(3 * H + u0 + r) * LSTM_NS + j] = dao[r];
        }
        if (live) {
            float* dp = dxg + o * 4 * H + u0;
            st4g(dp, make_float4(dai[0], dai[1], dai[2], dai[3]));
            st4g(dp + H, make_float4(daf[0], daf[1], daf[2], daf[3]));
            st4g(dp + 2 * H, make_float4(dag[0], dag[1], dag[2], dag[3]));
            st4g(dp + 3 * H, make_float4(dao[0], dao[1], dao[2], dao[3]));
        }
        __syncthreads();
        // dh_{t-1} += W_hh^T da_t   (this wave: its 16 hidden units x 16 sequences)
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[ks], das[(4 * ks + q) * LSTM_NS + j], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dhr[r] = acc[r];
        __syncthreads();                       // das is rewritten next step
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// FEW sequences (the recipes' batch sizes: 510 of them in DPRNN-TasNet's intra path at B = 2, 316 in GALRNet's): with 16 sequences per
// workgroup the sweep occupies nseq / 16 compute units and every step costs the full 16-column MFMA time of W_hh h (4H x H x 16
// MACs = 256 v_mfma_f32_16x16x4_f32 per SIMD at H = 128, ~8200 cycles) however few workgroups exist.  The variant below gives a
// workgroup FOUR sequences on v_mfma_f32_4x4x1_16b_f32 (16 independent 4 x 4 x 1 products per instruction, 8 cycles): the 64 rows a
// wave owns are the 16 blocks x 4 columns of B, the 4 sequences are the 4 rows of A (the same in every block), so a step costs a
// quarter of the MFMA cycles and four times as many compute units take part.  Lane l of a wave has two roles:
//   * ROW owner  (MFMA layout): row l of the wave's 64 = gate l / 16, unit 16 w + l % 16; the accumulator's four registers are the
//     four sequences.  Operand B = that row of W_hh, register-resident (H registers); operand A = h_{t-1}[sequence l % 4][k].
//   * CELL owner (update layout): (sequence l / 16, unit 16 w + l % 16) -- one cell per lane, so the transcendental work per lane is
//     5 evaluations per step instead of 20.  The two layouts are exchanged through 320 floats of wave-private LDS.
// Assumed operand layout of v_mfma_f32_4x4x1_16b_f32 (checked on the device by tools/mfma4x4_probe.hip before this path is enabled):
// lane l supplies A[block l / 4][row l % 4] and B[block l / 4][column l % 4] and receives D[block l / 4][row v][column l % 4] in
// register v (confirmed on the device by that probe, round 3: "layout: PASS").  Selection: few_sequences() below.
constexpr int NS4 = 4;
constexpr int XCH = 80;         // sequence stride of the layout-exchange scratch: (seq * 80 + part * 16 + unit) is conflict-free both ways

__device__ __forceinline__ f32x4 mfma4(const float a, const float b, const f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int H>
__global__ __launch_bounds__(H * 4) void lstm_fwd4_kernel(const float* __restrict__ xg, const float* __restrict__ whh,
                                                          float* __restrict__ hout, float* __restrict__ gates,
                                                          float* __restrict__ cstate, int nseq, int L, int dirs, int hrow) {
    constexpr int HS = H + 4;                  // row stride of the h panel: rows 16-byte aligned and on different bank groups
    const int dir = dirs == 2 ? (int)blockIdx.y : 0;
    const int reverse = dirs == 2 ? dir : dirs;
    {
        const size_t rows = (size_t)nseq * L;
        xg += dir * rows * 4 * H; whh += (size_t)dir * 4 * H * H; hout += hrow == H ? dir * rows * H : dir * H;
        if (gates) gates += dir * rows * 4 * H;
        if (cstate) cstate += dir * rows * H;
    }
    __shared__ __attribute__((aligned(16))) float hs[2][NS4 * HS];        // h_{t-1} / h_t as [sequence][unit]
    __shared__ float xch[H / 16][NS4 * XCH];                             // per wave: activated gates, [sequence][gate][unit of the wave]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ul = lane & 15, hi = lane >> 4, sj = lane & 3;
    const int u = 16 * w + ul;                 // hidden unit of this lane in both roles
    const int seq0 = blockIdx.x * NS4;

    float wf[H];                               // B operand: row (gate hi, unit u) of W_hh
    {
        const float* wr = whh + (size_t)(hi * H + u) * H;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 v = ld4g(wr + k);
            wf[k] = v.x; wf[k + 1] = v.y; wf[k + 2] = v.z; wf[k + 3] = v.w;
        }
    }
    for (int i = threadIdx.x; i < NS4 * HS; i += H * 4) hs[0][i] = 0.f;
    float c = 0.f;                             // cell owner: c of (sequence hi, unit u)
    __syncthreads();

    size_t xrow[NS4];                          // row owner: first row of each of the four sequences in xg / gates
    bool xlive[NS4];
#pragma unroll
    for (int v = 0; v < NS4; ++v) {
        xlive[v] = seq0 + v < nseq;
        xrow[v] = (size_t)(xlive[v] ? seq0 + v : nseq - 1) * L;
    }
    const bool live = seq0 + hi < nseq;        // cell owner
    const size_t crow = (size_t)(live ? seq0 + hi : nseq - 1) * L;
    // one evaluation serves both activations: r = 1 / (1 + 2^(kappa x)); sigmoid(x) = r with kappa = -log2 e, tanh(x) = 1 - 2 r with kappa = 2 log2 e
    const bool is_g = hi == 2;
    const float kappa = is_g ? 2.8853900817779268f : -1.4426950408889634f;

    auto load_x = [&](int t, float (&dst)[NS4]) {
#pragma unroll
        for (int v = 0; v < NS4; ++v) dst[v] = xg[(xrow[v] + t) * 4 * H + hi * H + u];
    };
    float xc[NS4], xn[NS4];
    load_x(reverse ? L - 1 : 0, xc);
    float* ex = xch[w];
    for (int step = 0; step < L; ++step) {
        const int t = reverse ? L - 1 - step : step;
        const int cur = step & 1;
        if (step + 1 < L) load_x(reverse ? t - 1 : t + 1, xn);          // next step's projection, in flight under the MFMAs
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;    // four chains: no MFMA waits for the one before it
        const float* hp = hs[cur] + sj * HS;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(hp + k);
            a0 = mfma4(hv.x, wf[k], a0);
            a1 = mfma4(hv.y, wf[k + 1], a1);
            a2 = mfma4(hv.z, wf[k + 2], a2);
            a3 = mfma4(hv.w, wf[k + 3], a3);
        }
        float act[NS4];                        // row owner: activated gate hi of unit u for the four sequences
#pragma unroll
        for (int v = 0; v < NS4; ++v) {
            const float pre = (a0[v] + a1[v]) + (a2[v] + a3[v]) + xc[v];
            const float r = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(kappa * pre));
            act[v] = is_g ? 1.f - 2.f * r : r;
            ex[v * XCH + hi * 16 + ul] = act[v];
        }
        if (gates) {
#pragma unroll
            for (int v = 0; v < NS4; ++v)
                if (xlive[v]) gates[(xrow[v] + t) * 4 * H + hi * H + u] = act[v];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the exchange is wave-private: LDS executes a wave's accesses in order
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // cell owner: the four gates of (sequence hi, unit u)
        const float gi = ex[hi * XCH + 0 * 16 + ul], gf = ex[hi * XCH + 1 * 16 + ul];
        const float gg = ex[hi * XCH + 2 * 16 + ul], go = ex[hi * XCH + 3 * 16 + ul];
        c = fmaf(gf, c, gi * gg);
        const float hn = go * tanh_f(c);
        hs[cur ^ 1][hi * HS + u] = hn;
        if (live) {
            hout[(crow + t) * hrow + u] = hn;
            if (cstate) cstate[(crow + t) * H + u] = c;
        }
#pragma unroll
        for (int v = 0; v < NS4; ++v) xc[v] = xn[v];
        __syncthreads();                       // h_t complete in hs[cur^1]; everyone is done reading hs[cur] and the exchange scratch
    }
}

template <int H>
__global__ __launch_bounds__(H * 4) void lstm_bwd4_kernel(const float* __restrict__ dhout, const float* __restrict__ gates,
                                                          const float* __restrict__ cstate, const float* __restrict__ whh,
                                                          float* __restrict__ dxg, int nseq, int L, int dirs, int hrow) {
    constexpr int HG = H + 16;                 // gate stride of the d(pre-activation) panel
    constexpr int DS = 4 * HG + 4;             // its sequence stride
    const int dir = dirs == 2 ? (int)blockIdx.y : 0;
    const int reverse = dirs == 2 ? dir : dirs;
    {
        const size_t rows = (size_t)nseq * L;
        dhout += hrow == H ? dir * rows * H : dir * H; gates += dir * rows * 4 * H; cstate += dir * rows * H; whh += (size_t)dir * 4 * H * H;
        dxg += dir * rows * 4 * H;
    }
    __shared__ __attribute__((aligned(16))) float das[NS4 * DS];         // d(pre-activation) of the current step as [sequence][gate][unit]
    __shared__ float xch[H / 16][NS4 * XCH];                             // per wave: the four partial sums of W_hh^T da, [sequence][part][unit]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ul = lane & 15, hi = lane >> 4, sj = lane & 3;
    const int u = 16 * w + ul;
    const int seq0 = blockIdx.x * NS4;

    // dh_{t-1}[seq][u] = sum over the 4H gate rows r of W_hh[r][u] da[seq][r].  The 16 blocks of the MFMA are (quarter hi of the
    // contraction = gate hi) x (unit quad): B operand = W_hh[hi * H + m][u], m = 0 .. H-1; the four quarters are summed afterwards.
    float wt[H];
#pragma unroll
    for (int m = 0; m < H; ++m) wt[m] = whh[(size_t)(hi * H + m) * H + u];
    const bool live = seq0 + hi < nseq;        // cell owner: (sequence hi, unit u)
    const size_t crow = (size_t)(live ? seq0 + hi : nseq - 1) * L;
    float dhr = 0.f, dc = 0.f;
    float* ex = xch[w];

    struct StepIn { float dh, i, f, g, o, cp; };
    auto load_step = [&](int t, StepIn& d) {
        const int tprev = reverse ? t + 1 : t - 1;
        const size_t o = crow + t;
        d.dh = dhout[o * hrow + u];
        const float* gp = gates + o * 4 * H + u;
        d.i = gp[0]; d.f = gp[H]; d.g = gp[2 * H]; d.o = gp[3 * H];
        d.cp = (tprev >= 0 && tprev < L) ? cstate[(crow + tprev) * H + u] : 0.f;
    };
    StepIn in, nx;
    const int t_first = reverse ? 0 : L - 1;
    load_step(t_first, in);
    float cc = cstate[(crow + t_first) * H + u];                         // c_t; the next step's c_t is this step's c_{t-1}
    for (int step = 0; step < L; ++step) {
        const int t = reverse ? step : L - 1 - step;                     // the forward sweep's LAST step first
        if (step + 1 < L) load_step(reverse ? t + 1 : t - 1, nx);        // in flight under this step's arithmetic
        const float dh = in.dh + dhr;
        const float tc = tanh_f(cc);
        const float dcc = fmaf(dh * in.o, 1.f - tc * tc, dc);
        const float dao = dh * tc * in.o * (1.f - in.o);
        const float dai = dcc * in.g * in.i * (1.f - in.i);
        const float daf = dcc * in.cp * in.f * (1.f - in.f);
        const float dag = dcc * in.i * (1.f - in.g * in.g);
        dc = dcc * in.f;
        float* dp = das + hi * DS + u;
        dp[0] = dai; dp[HG] = daf; dp[2 * HG] = dag; dp[3 * HG] = dao;
        if (live) {
            float* gp = dxg + (crow + t) * 4 * H + u;
            gp[0] = dai; gp[H] = daf; gp[2 * H] = dag; gp[3 * H] = dao;
        }
        __syncthreads();
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        const float* ap = das + sj * DS + hi * HG;                       // A operand: da[sequence l % 4][gate hi][m]
#pragma unroll
        for (int m = 0; m < H; m += 4) {
            const float4 dv = *reinterpret_cast<const float4*>(ap + m);
            a0 = mfma4(dv.x, wt[m], a0);
            a1 = mfma4(dv.y, wt[m + 1], a1);
            a2 = mfma4(dv.z, wt[m + 2], a2);
            a3 = mfma4(dv.w, wt[m + 3], a3);
        }
#pragma unroll
        for (int v = 0; v < NS4; ++v) ex[v * XCH + hi * 16 + ul] = (a0[v] + a1[v]) + (a2[v] + a3[v]);   // quarter hi of (sequence v, unit u)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dhr = (ex[hi * XCH + ul] + ex[hi * XCH + 16 + ul]) + (ex[hi * XCH + 32 + ul] + ex[hi * XCH + 48 + ul]);   // cell owner again
        cc = in.cp;
        in = nx;
        __syncthreads();                       // das and the exchange scratch are rewritten next step
    }
}

// Which sweep kernel: four sequences per workgroup while those workgroups still fit ONE round on the 256 compute units, else sixteen.
// Measured on MI355X (tools/lstm_bench.py, bi-directional fwd + bwd, H = 128, profiles/r03a_lstm4.txt): 510 sequences x 250 steps
// (DPRNN-TasNet's inter-chunk path at B = 2: 256 four-sequence workgroups against 64 sixteen-sequence ones) 3.74 ms against 5.45;
// 2040 sequences (1020 workgroups = four rounds against one) 11.85 ms against 10.73.  SEPK_LSTM_NS4 = 0 / 1 forces sixteen / four.
// The caller can force either one per call: bits 8-9 of `reverse` (SEP_LSTM_FORCE16 / SEP_LSTM_FORCE4, include/sepkernels.h) -- the tests
// run every case through both.
bool few_sequences(int nseq, int reverse, int force) {
    static const int mode = getenv("SEPK_LSTM_NS4") ? atoi(getenv("SEPK_LSTM_NS4")) : 2;
    if (force == 1 || force == 2) return force == 2;
    if (mode == 0 || mode == 1) return mode == 1;
    return ((nseq + NS4 - 1) / NS4) * (reverse == 2 ? 2 : 1) <= 256;
}

template <int H>
int launch_fwd(const float* xg, const float* whh, float* hout, float* gates, float* cstate, int nseq, int L, int reverse, int force, int hs, hipStream_t st) {
    if (few_sequences(nseq, reverse, force)) {
        hipLaunchKernelGGL((lstm_fwd4_kernel<H>), dim3((nseq + NS4 - 1) / NS4, reverse == 2 ? 2 : 1), dim3(H * 4), 0, st, xg, whh, hout, gates, cstate, nseq, L, reverse, hs);
        return 0;
    }
    hipLaunchKernelGGL((lstm_fwd_kernel<H>), dim3((nseq + LSTM_NS - 1) / LSTM_NS, reverse == 2 ? 2 : 1), dim3(H * 4), 0, st, xg, whh, hout, gates, cstate, nseq, L, reverse, hs);
    return 0;
}
template <int H>
int launch_bwd(const float* dhout, const float* gates, const float* cstate, const float* whh, float* dxg, int nseq, int L, int reverse, int force, int hs, hipStream_t st) {
    if (few_sequences(nseq, reverse, force)) {
        hipLaunchKernelGGL((lstm_bwd4_kernel<H>), dim3((nseq + NS4 - 1) / NS4, reverse == 2 ? 2 : 1), dim3(H * 4), 0, st, dhout, gates, cstate, whh, dxg, nseq, L, reverse, hs);
        return 0;
    }
    hipLaunchKernelGGL((lstm_bwd_kernel<H>), dim3((nseq + LSTM_NS - 1) / LSTM_NS, reverse == 2 ? 2 : 1), dim3(H * 4), 0, st, dhout, gates, cstate, whh, dxg, nseq, L, reverse, hs);
    return 0;
}

}  // namespace

extern "C" int sep_lstm_fwd(const float* xg, const float* w_hh, float* h_out, float* gates, float* cstate, int nseq, int L,
                            int H, int reverse, sep_stream_t stream) {
    SEP_REQUIRE(xg && w_hh && h_out, "sep_lstm_fwd: null pointer");
    const int force = (reverse >> 8) & 3;
    const bool interleaved = (reverse & SEP_LSTM_INTERLEAVED) != 0;
    reverse &= 0xff;
    SEP_REQUIRE(nseq > 0 && L > 0 && reverse >= 0 && reverse <= 2 && force <= 2, "sep_lstm_fwd: bad sizes / direction");
    SEP_REQUIRE(!interleaved || reverse == 2, "sep_lstm_fwd: SEP_LSTM_INTERLEAVED goes with both directions in one call (reverse = 2)");
    const int hs = interleaved ? 2 * H : H;
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 16: launch_fwd<16>(xg, w_hh, h_out, gates, cstate, nseq, L, reverse, force, hs, st); break;
        case 32: launch_fwd<32>(xg, w_hh, h_out, gates, cstate, nseq, L, reverse, force, hs, st); break;
        case 64: launch_fwd<64>(xg, w_hh, h_out, gates, cstate, nseq, L, reverse, force, hs, st); break;
        case 128: launch_fwd<128>(xg, w_hh, h_out, gates, cstate, nseq, L, reverse, force, hs, st); break;
        default: SEP_REQUIRE(false, "sep_lstm_fwd: hidden size %d not supported (16, 32, 64, 128)", H);
    }
    SEP_CHECK_LAUNCH("sep_lstm_fwd");
    return 0;
}

extern "C" int sep_lstm_bwd(const float* dh_out, const float* gates, const float* cstate, const float* w_hh, float* dxg,
                            int nseq, int L, int H, int reverse, sep_stream_t stream) {
    SEP_REQUIRE(dh_out && gates && cstate && w_hh && dxg, "sep_lstm_bwd: null pointer");
    const int force = (reverse >> 8) & 3;
    const bool interleaved = (reverse & SEP_LSTM_INTERLEAVED) != 0;
    reverse &= 0xff;
    SEP_REQUIRE(nseq > 0 && L > 0 && reverse >= 0 && reverse <= 2 && force <= 2, "sep_lstm_bwd: bad sizes / direction");
    SEP_REQUIRE(!interleaved || reverse == 2, "sep_lstm_bwd: SEP_LSTM_INTERLEAVED goes with both directions in one call (reverse = 2)");
    const int hs = interleaved ? 2 * H : H;
    hipStream_t st = (hipStream_t)stream;
    switch (H) {
        case 16: launch_bwd<16>(dh_out, gates, cstate, w_hh, dxg, nseq, L, reverse, force, hs, st); break;
        case 32: launch_bwd<32>(dh_out, gates, cstate, w_hh, dxg, nseq, L, reverse, force, hs, st); break;
        case 64: launch_bwd<64>(dh_out, gates, cstate, w_hh, dxg, nseq, L, reverse, force, hs, st); break;
        case 128: launch_bwd<128>(dh_out, gates, cstate, w_hh, dxg, nseq, L, reverse, force, hs, st); break;
        default: SEP_REQUIRE(false, "sep_lstm_bwd: hidden size %d not supported (16, 32, 64, 128)", H);
    }
    SEP_CHECK_LAUNCH("sep_lstm_bwd");
    return 0;
}
