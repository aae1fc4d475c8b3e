/*
 * sepkernels.h -- C ABI of libsepkernels.so: the MI355X (gfx950) kernels behind the
 * Conv-TasNet separation path of tky823/DNN-based_source_separation.
 *
 * The reference is pure PyTorch and has NO FFI for this path; each entry
 * point below therefore names the reference *Python* interface whose arithmetic it replaces
 * (file:line under /root/reference/src).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference adds to route those interfaces here.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types; every pointer is a DEVICE pointer
 *     unless the name ends in _host.
 *   - activations are fp32, layout (batch, channel, frame) as in the reference, frame
 *     contiguous, with a padded row stride `ldt` (floats, multiple of 128).  Frames
 *     [T, ldt) of every tensor a kernel WRITES are set to 0.
 *   - gLN statistics are carried as double[B][SEP_STATS_SLOTS][2]: SEP_STATS_SLOTS independent
 *     {sum, sum of squares} accumulators over the valid (C, T) region of a sample (producer blocks
 *     spread their fp64 atomics over the slots; consumers add the slots up).  The caller zeroes them.
 *   - the caller owns every buffer (inputs, outputs, saved activations, workspaces).
 *     The library keeps no global mutable state, is re-entrant, never synchronises the
 *     device and launches only on the stream it is given (the caller selects the device).
 *   - return 0 on success, <0 on error; sep_last_error() gives the thread-local message.
 */
#ifndef SEPKERNELS_H
#define SEPKERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sep_stream_t; /* hipStream_t */

#define SEP_ABI_VERSION 23
#define SEP_STATS_SLOTS 16
#define SEP_ARRIVE_INTS 17 /* arrival counters per sample: one per slot + one for the slots (csrc/common.hpp, gln_bwd_publish) */

int sep_version(void);
const char* sep_last_error(void);

/* ---- prologue modes applied to the X operand while it is staged into LDS ---------- */
#define SEP_PRO_NONE 0
#define SEP_PRO_PRELU 1     /* x -> PReLU(x; alpha)                         conv_tasnet.py:373 */
#define SEP_PRO_GLN 2       /* x -> gLN(x)                                  modules/norm.py:27 */
#define SEP_PRO_GLN_PRELU 3 /* x -> gLN(PReLU(x; alpha))                    tdcn.py:113-116,182-186 */
#define SEP_PRO_GLN_BWD 4   /* X = d(gLN out); uses pro_aux = pre-activation a:
                               da = rstd*(gamma*X - mg - xhat*mgx) * PReLU'(a); da is also stored to pro_store
                               and sum(du * a * [a<=0]) is accumulated into pro_dalpha.  pro_store may be X itself (in place)
                               only when M <= 128: with several 128-row tiles the others still read the untouched X */

/* ---- epilogue flags ----------------------------------------------------------------- */
#define SEP_EPI_STATS_PRELU 1 /* accumulate sum/sumsq of PReLU(y; epi_alpha) into epi_stats (y itself is stored) */
#define SEP_EPI_RESIDUAL 2    /* y += epi_res                                tdcn.py:144-145 */
#define SEP_EPI_SIGMOID 4     /* y = 1/(1+exp(-y))                           conv_tasnet.py:375 */
#define SEP_EPI_PRELU_BWD 8   /* y = y * PReLU'(epi_aux); epi_dalpha += sum(y_in * epi_aux * [epi_aux<=0]) */
#define SEP_EPI_ROWSUMS 16    /* epi_rowpart[b][m][t/64][0..1] = sum_t y, sum_t y*u, u = epi_aux (or PReLU(epi_aux)) */
#define SEP_EPI_ROWSUMS_PRELU 32

/* ---- arithmetic of the contraction (fp32 operands and fp32 accumulation in both) ------------------------------
 * F32:    v_mfma_f32_32x32x2_f32 on the operands as they are.
 * BF16X6: every operand value is split EXACTLY into three truncated-bf16 parts (8+8+8 significand bits) and
 *         x*y = hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (the three dropped part
 *         products are <= 2^-24 |xy| each); results agree with fp64 as closely as the F32 path's do (tests), at
 *         6 x 32 instead of 8 x 64 matrix-pipe cycles per 16-deep chunk.  Inputs are assumed finite.
 * F16X3:  sep_pw_gemm only (sep_pw_wgrad treats it as BF16X6).  Two-part fp16 split x*2^s = hi + lo (11 + 11 bits),
 *         x*y = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (3 x 32 cycles), operands brought into fp16's exponent
 *         range by exact power-of-two scales: one for A from the caller's bound a_amax >= max|A| (device scalar), one per
 *         column of X chosen and adjusted inside the kernel.  The error is relative to |A||X| per output (like fp32
 *         accumulation's), not elementwise: values ~2^-25 below their column's maximum lose low bits.  Without a_amax
 *         the call runs as BF16X6.
 *         PACKED WEIGHTS (A_pk != NULL, the form the Conv-TasNet step uses): A has been split ONCE per pass by
 *         sep_pack_weights into {hi, lo} fp16 pairs with one power-of-two scale PER ROW of A (a_rscale[m] undoes it in the
 *         epilogue); the kernel then DMAs MFMA-ready A operands, and the X tile is put through the prologue and split once
 *         per workgroup (cooperatively, shared through LDS) instead of once per wave.  Needs M % 128 == 0. */
#define SEP_ARITH_F32 0
#define SEP_ARITH_BF16X6 1
#define SEP_ARITH_F16X3 2

/* Pointwise (1x1) convolution as a GEMM on MFMA (fp32 in / fp32 accumulate):
 *     Y[b][m][t] = epilogue( sum_k A[m][k] * prologue(X[b][k][t]) + bias[m] )
 * Replaces nn.Conv1d(kernel_size=1) at tdcn.py:86,173,175 and conv_tasnet.py:335,341 in the
 * forward direction, and the same layers' input-gradient (A transposed) in backward.
 * M and K must be multiples of 16 (K) / any (M); ldt multiple of 128. */
typedef struct sep_gemm_desc {
    int32_t B, M, K, T, ldt;
    int32_t trans_a;    /* 0: A is [M][K] row-major ; 1: A is [K][M] row-major (input-gradient form) */
    int32_t k_split;    /* 0, or multiple of 16: contraction rows k >= k_split are read from (A2, X2) at row k-k_split */
    int32_t m_split;    /* 0, or multiple of 128: output rows m >= m_split are written to Y2 (row m-m_split) */
    int32_t pro_mode;   /* SEP_PRO_* */
    int32_t epi_flags;  /* SEP_EPI_* bitmask; RESIDUAL applies to Y rows only */
    int32_t accumulate; /* 1: Y2 (or Y when m_split==0) += result instead of = */
    int32_t arith;      /* SEP_ARITH_* */
    float eps;
    double count; /* number of valid elements per sample (C*T) of the gLN used by the prologue */
    const float* A;
    const float* A2;
    const float* X;
    const float* X2;
    float* Y;
    float* Y2;
    const float* bias; /* [M] or NULL */
    const float* pro_alpha;
    const double* pro_stats;
    const float* pro_gamma;
    const float* pro_beta;
    const float* pro_aux;
    const float* pro_bsum; /* GLN_BWD: [B][2] = mean(gamma g), mean(gamma g xhat) of the gLN being back-propagated (bsum1 of sep_dwconv_bwd,
                              bsum of sep_gln_bwd_from_wgrad / sep_gln_bwd_finalize) */
    const double* pro_bacc; /* GLN_BWD, instead of pro_bsum: [B][SEP_STATS_SLOTS][2], the raw sums {sum_c gamma_c sum_t g, sum_c gamma_c sum_t g*u} a
                               producer only added up (sep_dwconv_bwd's bacc1 without arrive1 / bsum1); the kernel forms the two means itself */
    float* pro_store;
    double* pro_dalpha;
    const float* epi_alpha;
    double* epi_stats;
    const float* epi_res;
    const float* epi_aux;
    double* epi_dalpha;
    float* epi_rowpart;
    const float* a_amax; /* SEP_ARITH_F16X3: device scalar >= max|A| (and |A2|); any upper bound, e.g. over all parameters */
    const void* A_pk;       /* SEP_ARITH_F16X3: A ([M][K], already in the orientation of the product: the transpose and the
                               [A|A2] concatenation of a k_split call are done by the packer) as written by sep_pack_weights,
                               or NULL.  A / A2 / trans_a must still describe the fp32 weights: shapes the packed kernel does
                               not cover run on them. */
    const float* a_rscale;  /* [M] = 2^-e_m, the inverse row scales belonging to A_pk */
} sep_gemm_desc;

int sep_pw_gemm(const sep_gemm_desc* d, sep_stream_t stream);

/* Weight packer of the SEP_ARITH_F16X3 path: for every segment, A[m][k] = trans ? W[k*ldw + m] : W[m*ldw + k] (M x K,
 * K % 8 == 0), e_m = 13 - exponent(max_k |A[m][k]|), and each group of 8 consecutive k becomes 32 bytes
 *     dst[(m*K/8 + g)*32 ..] = { fp16 hi[8] , fp16 lo[8] },  A*2^e_m = hi + lo (hi toward zero, 11 + 11 significand bits)
 * i.e. 4 bytes per weight like the fp32 matrix; rscale[m] = 2^-e_m.  One launch for up to 256 segments (all 1x1-conv
 * weights of a Conv-TasNet in both orientations).  Replaces nothing in the reference: it is the once-per-pass half of the
 * operand split of the nn.Conv1d(kernel_size=1) products (tdcn.py:86,173,175; conv_tasnet.py:335,341). */
typedef struct sep_pack_seg {
    const float* W;
    void* dst;      /* M*K*4 bytes, 32-byte aligned */
    float* rscale;  /* [M] */
    int32_t M, K, trans, ldw;
} sep_pack_seg;
int sep_pack_weights(const sep_pack_seg* segs_host, int nseg, sep_stream_t stream);

/* Weight gradient of a pointwise convolution (reduction over batch and frames) on MFMA:
 *     partial[s][m][n] = sum over the (b,t) columns of slab s of  Gp[b][m][t] * Xp[b][n][t]
 *     partial_bias[s][m] = sum of Gp[b][m][t]
 * followed by sep_reduce_slabs.  Replaces autograd's conv weight/bias gradient for the same
 * nn.Conv1d layers, and (with unfolded frames as X) for Encoder.conv1d (filterbank.py:212)
 * and Decoder.conv_transpose1d (filterbank.py:243). */
typedef struct sep_wgrad_desc {
    int32_t B, M, N, T, ldt;
    int32_t g_split; /* 0, or multiple of 128: rows m >= g_split of G are read from G2 */
    int32_t g_mul;   /* 1: Gp = G * Gaux[b / g_div] (latent = mask * w) */
    int32_t g_div;
    int32_t x_mode; /* SEP_PRO_NONE / PRELU / GLN / GLN_PRELU */
    int32_t x_div;  /* X (and its gLN stats) are indexed with b / x_div */
    int32_t nsplit; /* number of partial slabs, <= B*ldt/32; slab s covers the chunk range [s*ceil(C/nsplit), (s+1)*ceil(C/nsplit)) of the C = B*ldt/q
                      * frame chunks (q = 16 or 32, the kernel's choice).  SAMPLE-ALIGNED slabs: with nsplit = B*k and k a divisor of ldt/32,
                      * slab s holds frames of sample s/k only, in every kernel */
    int32_t arith;  /* SEP_ARITH_* */
    float eps;
    double count;
    const float* G;
    const float* G2;
    const float* Gaux;
    const float* X;
    const float* x_alpha;
    const double* x_stats;
    const float* x_gamma;
    const float* x_beta;
    float* partial;      /* [nsplit][M][N] */
    float* partial_bias; /* [nsplit][M] or NULL */
    /* ABI 23: the second G source PRE-SPLIT by sep_split_rows (NULL: not used).  With M = 256, g_split = 128 and slabs that lie inside one sample the fp16
     * weight-gradient kernel takes rows m >= g_split as ready {hi, lo} operands (no split arithmetic for them in any layer that reads the same tensor:
     * the skip gradient dS of the Conv-TasNet step is the second source of all 24 heads' weight gradients); G2 must still be given (other kernels,
     * other shapes fall back to it). */
    const void* G2_pre;    /* [B][M - g_split][ldt / 32] lines of 128 bytes: 32 hi then 32 lo fp16 of x * 2^e */
    const int32_t* g2_exps; /* [B][M - g_split]: e */
    const float* g2_sums;  /* [nsplit][M - g_split]: sum over the slab's frames (the bias partials of those rows), or NULL with partial_bias == NULL */
} sep_wgrad_desc;

int sep_pw_wgrad(const sep_wgrad_desc* d, sep_stream_t stream);
/* n <= 8 weight gradients that agree in EVERYTHING but G, G2, X, partial, partial_bias (e.g. the conv1 weight gradients of consecutive TCN layers,
 * which nothing in the backward pass waits for), as ONE launch where the fp16 producer / consumer kernel takes the shape: the grid stays one
 * workgroup per compute unit, so the caller gives every product 1/n of the slabs it would give a single call (nsplit of the descriptors) and
 * gets n times longer contractions per workgroup and 1/n of the slab traffic.  Other shapes / arithmetics: n calls of sep_pw_wgrad.
 * Same results as n separate calls with the same nsplit.  (ABI 23; reference: the same convolutions' weight gradients, tdcn.py:86.) */
int sep_pw_wgrad_batch(const sep_wgrad_desc* descs_host, int n, sep_stream_t stream);
/* Rows of an activation tensor split ONCE for every weight gradient that takes them as (second) G operand: for row r of sample b
 *   e = 13 - exponent(max_t |x[b][r][t]|)   (the row's maximum lands in [2^12, 2^13); an all-zero row: e = 0)
 *   out line (b, r, p) = { hi[32] | lo[32] } fp16 of x[b][r][32 p .. 32 p + 31] * 2^e,  hi = fp16 toward zero, lo = fp16(x 2^e - hi)
 *   sums[(b * k + q) * C + r] = sum of x[b][r][t] over the q-th of k equal frame ranges of the row (k slabs per sample: the bias partials)
 * out has the shape and row pitch of x (B, C, ldt floats).  ldt % (32 k) == 0.  (ABI 23; no reference counterpart: it is half of the operand split
 * of the heads' weight gradient, reference src/models/tdcn.py:173,175, hoisted out of the 24 layers.) */
int sep_split_rows(const float* x, void* out, int32_t* exps, float* sums, int B, int C, int T, int ldt, int k, sep_stream_t stream);

/* dst[i] (+)= scale * sum_s src[s*stride + i] for up to 64 independent segments in one launch
 * (deterministic second stage of every split reduction). */
typedef struct sep_reduce_seg {
    const float* src;
    float* dst;
    int32_t n, nslab;
    int64_t stride;
    int32_t accumulate;
    float scale;
} sep_reduce_seg;
int sep_reduce_slabs(const sep_reduce_seg* segs_host, int nseg, sep_stream_t stream);

/* dst(double scalar partials) -> float parameter gradient: dst[i] (+)= (float)src[i] */
int sep_f64_to_f32(const double* src, float* dst, int n, int accumulate, sep_stream_t stream);

/* Encoder.forward (filterbank.py:222-230) with ConvTasNet's input padding (conv_tasnet.py:145-149):
 *   w[b][n][f] = sum_{c,k} E[n][c][k] * xpad[b][c][S*f+k] (+ReLU), xpad = x shifted right by pad_left, zero elsewhere.
 * Also accumulates the gLN statistics of w (separator.norm1d, conv_tasnet.py:370). */
int sep_encoder_fwd(const float* x, const float* E, float* w, double* stats, int B, int Cin, int Tin, int N, int L,
                    int S, int F, int ldt, int pad_left, int relu, sep_stream_t stream);

/* frames[b][c*L+k][f] = xpad[b][c][S*f+k]  (the im2col operand of the encoder/decoder weight gradients) */
int sep_unfold(const float* x, float* frames, int Bp, int C, int Tin, int L, int S, int F, int ldt, int pad_left,
               sep_stream_t stream);

/* Depthwise dilated conv of ResidualBlock1d/DepthwiseSeparableConv1d (tdcn.py:113-132,177-186), forward:
 *   v = gLN(PReLU(a; alpha1)) ; v = 0 outside [0,T) ; z[c][t] = bd[c] + sum_k wd[c][k] * v[c][t+(k-1)*d]
 * and accumulation of the statistics of PReLU(z; alpha2) for the following gLN.  P = 3 only. */
int sep_dwconv_fwd(const float* a, const double* stats1, const float* gamma1, const float* beta1, const float* alpha1,
                   const float* wd, const float* bd, const float* alpha2, float* z, double* stats2, int B, int C, int T,
                   int ldt, int dilation, float eps, sep_stream_t stream);

/* Backward of [gLN2 o PReLU2 o depthwise] given dv2 = d(gLN2 output):
 *   du2 = r2*(gamma2*dv2 - mg2 - xhat2*mgx2) ; dz = du2*PReLU'(z) ; dv1 = depthwise^T(dz)
 * bsum2 [B][2] = {mg2, mgx2} = mean(gamma2 dv2), mean(gamma2 dv2 xhat2), written by the producer of dv2's sums (sep_gln_bwd_from_wgrad).
 * For gLN1 this kernel is the producer: bacc1 [B][SEP_STATS_SLOTS][2] (fp64, zeroed by the caller) receives its workgroups'
 * {sum_c gamma1_c sum_t dv1, sum_c gamma1_c sum_t dv1*u1}, arrive1 [B][SEP_ARRIVE_INTS] (int, zeroed) counts them, and the sample's LAST workgroup stores
 * bsum1 [B][2] = {mean(gamma1 dv1), mean(gamma1 dv1 xhat1)} for the consumer's prologue (sep_gemm_desc.pro_bsum) -- no second-stage launch
 * between the two (round 2: sep_gln_bwd_finalize, 98 launches per step on the critical path).  arrive1 = bsum1 = NULL: the sums only (the
 * consumer forms the means from them: sep_gemm_desc.pro_bacc -- what the Conv-TasNet step uses, this kernel has thousands of short
 * workgroups and the arrival protocol's two waited-for round trips cost it 10 us per launch); all three NULL: none of it.
 * Writes dv1 and, per (b, c, 1024-frame tile), 8 partial row sums into rowpart[b][c][ntile][8]:
 *   {sum dv1, sum dv1*u1, sum dz, sum dz*v1[t-d], sum dz*v1[t], sum dz*v1[t+d], sum du2*z*[z<=0], 0}
 * (u1 = PReLU(a), v1 = gLN1(u1) inside [0,T) and 0 outside; ntile = ceil(ldt/1024)).
 * bd (the depthwise bias, may be NULL): with it the kernel forms z = bd + depthwise(v1) again from `a`, which it reads anyway, instead of
 * reading z -- three streams of HBM instead of four (rows of up to 8192 frames; longer rows and bd = NULL read z). */
int sep_dwconv_bwd(const float* dv2, const float* z, const float* a, const double* stats1, const float* gamma1,
                   const float* beta1, const float* alpha1, const double* stats2, const float* gamma2,
                   const float* alpha2, const float* bsum2, const float* wd, const float* bd, float* dv1, float* rowpart,
                   double* bacc1, int* arrive1, float* bsum1, int B, int C, int T, int ldt, int dilation, float eps, sep_stream_t stream);

/* Second stage of every gLN backward (Appendix A of SURVEY.md).  rowpart is [B][C][ntile][nq], nq in {2, 8}:
 *   R1 = sum_tiles rowpart[..][0], R2 = sum_tiles rowpart[..][1]
 *   pbeta[b][c] = R1 ; pgamma[b][c] = r_b*(R2 - mu_b*R1)
 *   bsum[b] = { sum_c gamma_c*R1 / count , sum_c gamma_c*pgamma[b][c] / count }     (bsum may be NULL: inside the TCN layers of the fused
 *             Conv-TasNet path the producers publish these means themselves -- sep_dwconv_bwd's bsum1, sep_gln_bwd_from_wgrad's bsum -- and
 *             this kernel only forms the parameter gradients, off the critical path)
 *   nq == 8 additionally (pextra holds B*C*4 + B + B*C floats; the last B*C are scratch):
 *     pextra[b*4C + c]           = sum_tiles rowpart[..][2]      (depthwise bias gradient, per sample)
 *     pextra[b*4C + C + 3c + k]  = sum_tiles rowpart[..][3+k]    (depthwise weight gradient [C][3], per sample)
 *     pextra[B*4C + b]           = sum_{c,tiles} rowpart[..][6]  (PReLU slope gradient, per sample) */
int sep_gln_bwd_finalize(const float* rowpart, int ntile, int nq, const double* stats, const float* gamma, double count,
                         float eps, float* bsum, float* pbeta, float* pgamma, float* pextra, int B, int C,
                         sep_stream_t stream);

/* sep_gln_bwd_finalize for up to 64 gLNs in one launch per stage (same outputs; bsum may be NULL per segment). */
typedef struct sep_finalize_seg {
    const float* rowpart;
    const double* stats;
    const float* gamma;
    float* bsum;
    float* pbeta;
    float* pgamma;
    float* pextra;
    double count;
    float eps;
    int32_t ntile, nq, B, C;
} sep_finalize_seg;
int sep_gln_bwd_finalize_batch(const sep_finalize_seg* segs_host, int nseg, sep_stream_t stream);

/* gLN backward statistics FROM the weight gradient (round 3).  For y = W v with v = gLN(u), u = PReLU(z): the gradient at v is
 * dv = W^T g, and the row sums its gLN backward needs are contractions the weight gradient has already done,
 *     R1[n] = sum_t dv[n][t] = sum_m W[m][n] gs[m],        R2[n] = sum_t dv[n][t] u[n][t] = sum_m W[m][n] raw[m][n],
 * raw = sum_t g[m][t] u[n][t] (sep_pw_wgrad with x_mode = SEP_PRO_PRELU: the gain and shift of the norm left out), gs = sum_t g (its
 * bias slabs), both per sample -- the slabs must be SAMPLE-ALIGNED: nsplit = B * slabs_per_sample with slabs_per_sample dividing ldt / 32,
 * so that slab s holds frames of sample s / slabs_per_sample only.  Then the input-gradient product W^T g needs no ROWSUMS epilogue
 * and does not read z (reference tdcn.py:173-175 backward + modules/norm.py:18,27 backward).  Outputs:
 *     dW_b[b][m][n] = sc_bn raw_b[m][n] + sh_bn gs_b[m]      (the true per-sample weight gradient; sum over b with sep_reduce_slabs)
 *     pbeta[b][n] (+)= R1,  pgamma[b][n] (+)= rstd_b (R2 - mu_b R1)      (accumulate = 1 adds: a second product feeding the same gLN)
 *     bacc[b][slot] += { sum_n gamma_n R1, sum_n gamma_n R2 }   (fp64, zeroed by the caller; `products` calls feed one gLN -- e.g. the
 *                          output and the skip product when their weights are not adjacent -- each adding its share)
 *     bsum[b] = { mean(gamma dv), mean(gamma dv xhat) }         stored by the LAST workgroup of the last call (arrive [B][SEP_ARRIVE_INTS], int,
 *                          zeroed by the caller, counts the arrivals): what the consumer of dv reads (sep_dwconv_bwd's bsum2)
 * W is [M][N] row-major (rows of adjacent matrices may be handed over as one, e.g. [Wo; Ws]). */
int sep_gln_bwd_from_wgrad(const float* part, const float* part_bias, const float* W, const double* stats, const float* gamma,
                           const float* beta, double count, float eps, float* dW_b, float* pbeta, float* pgamma, double* bacc,
                           int* arrive, float* bsum, int B, int M, int N, int slabs_per_sample, int accumulate, int products,
                           sep_stream_t stream);

/* Backward tail of the separator head: dw = r0*(gamma0*dvw - mg - xhat*mgx) + dwm, times [w>0] if the encoder has ReLU
 * (bsum0 [B][2] = {mg, mgx} from sep_gln_bwd_finalize).
 * In place on dvw. (conv_tasnet.py:370 gLN backward + conv_tasnet.py:159-160 product rule + filterbank.py:227) */
int sep_head_bwd(float* dvw, const float* w, const float* dwm, const double* stats0, const float* gamma0,
                 const float* bsum0, int B, int C, int T, int ldt, double count, float eps, int relu,
                 sep_stream_t stream);

/* mask * w, Decoder.forward = basis synthesis + overlap-add, and the crop (conv_tasnet.py:159-169, filterbank.py:245-247):
 *   est[b][s][c][tau] = sum_{n,f,k: S*f+k = tau+pad_left} w[b][n][f] * m[b][s][n][f] * D[n][c][k]
 * latent (may be NULL) receives w*m as (B, n_src, N, ldt). */
int sep_decoder_fwd(const float* w, const float* m, const float* D, float* est, float* latent, int B, int n_src, int N,
                    int Cout, int L, int S, int F, int ldt, int Tout, int pad_left, sep_stream_t stream);

/* Backward of the same: given d_est, writes dpre as (B, n_src*N, ldt) and dwm[b][n][f] = sum_s dlatent*m.
 * raw_mask = 0: dpre = d(mask pre-activation) of a sigmoid mask; 1: dpre = d(mask) (softmax mask: sep_softmax_ch_bwd next). */
int sep_decoder_bwd(const float* d_est, const float* w, const float* m, const float* D, float* dpre, float* dwm, int B,
                    int n_src, int N, int Cout, int L, int S, int F, int ldt, int Tout, int pad_left, int raw_mask,
                    sep_stream_t stream);

/* mask_nonlinear = 'softmax' (conv_tasnet.py:353-357, 375): nn.Softmax(dim=1) over the n_src*N channels of a frame, in place
 * on y (B, C, ldt); backward in place on g given the forward output y: g <- y * (g - sum_c g*y).  Frames >= T are zeroed. */
int sep_softmax_ch_fwd(float* y, int B, int C, int T, int ldt, sep_stream_t stream);
int sep_softmax_ch_bwd(const float* y, float* g, int B, int C, int T, int ldt, sep_stream_t stream);

/* Cumulative layer norm (causal cLN), CumulativeLayerNorm1d of reference src/modules/norm.py:42-101 and its autograd backward:
 *   y = (x - m_t) / (sqrt(v_t) + eps) * gamma_c + beta_c with the mean / biased variance of all channels and frames <= t.
 * x, y, dy, dx: (B, C, ldt) fp32, frames contiguous, ldt % 4 == 0, frames >= T written as zeros; mean, rstd: (B, ldt) fp32 (ABI 20: rows of
 * ldt, entries >= T untouched), written by the forward and read by the backward; ws: scratch of sep_cln_ws_bytes(B, C, T, ldt) bytes, 16-byte
 * aligned (ABI 21: for C <= 512 one pass each way -- a workgroup holds all channels of a 32-frame tile and takes the prefix / suffix of the
 * earlier / later tiles from a look-back chain whose records live in ws, with the per-tile partial sums of the parameter gradients behind
 * them; for wider rows the column sums of three launches: (B, 2, ldt) fp64);
 * dgamma_part, dbeta_part: (B, C) per-sample sums, to be added over the samples (sep_reduce_slabs).
 * alpha (ABI 20; may be NULL): the single slope of a PReLU in FRONT of the norm (tdcn.py:113-116, 182-186: nonlinear1d then norm1d): the
 * kernels normalise u = PReLU(x; alpha), dx is the gradient at x, and dalpha_part (B, C) receives sum_t du * x * [x <= 0] per row. */
size_t sep_cln_ws_bytes(int B, int C, int T, int ldt);
int sep_cln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, double* ws, int B, int C,
                int T, int ldt, float eps, const float* alpha, sep_stream_t stream);
int sep_cln_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                float* dgamma_part, float* dbeta_part, double* ws, int B, int C, int T, int ldt, float eps, const float* alpha,
                float* dalpha_part, sep_stream_t stream);

/* gLN on TOKEN-MAJOR rows (ABI 20): x, y, dy, dx (nseq, L, C) with the C features contiguous -- the layout between the attention / LSTM /
 * Linear layers of the dual-path separators (dptnet.py:505-560 `norm1d(x.permute(1, 2, 0))`): statistics over the L*C values of a sequence
 * (biased variance, eps inside the root, as nn.GroupNorm(1, C)), gain / shift per feature.  C must divide 1024, C >= 4, (L*C) % 4 == 0.
 * stats (nseq, 2) = {mean, rstd}, written forward, read backward; part (nseq, 2, C) = {sum_t dy*xhat | sum_t dy} per sequence: summed over
 * the sequences they are d(gamma), d(beta).  ws (ABI 21): scratch of sep_gln_tokens_ws_bytes(nseq, L, C) bytes, 8-byte aligned; 0 bytes (ws may
 * be NULL) while the sequences alone fill the chip -- few long sequences (GALRNet normalises one per sample) are cut into slices whose
 * partial sums meet there. */
size_t sep_gln_tokens_ws_bytes(int nseq, int L, int C);
int sep_gln_tokens_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, void* ws, int nseq, int L, int C, float eps,
                       sep_stream_t stream);
int sep_gln_tokens_bwd(const float* dy, const float* x, const float* gamma, const float* stats, float* dx, float* part, void* ws, int nseq, int L,
                       int C, sep_stream_t stream);

/* Scaled dot-product attention of the dual-path separators' transformer blocks (ABI 21): O = dropout(softmax(scale Q K^T)) V per (sequence,
 * head) -- the core of nn.MultiheadAttention as the reference calls it (models/dptnet.py:505-527, models/galr.py:160-226, the
 * nn.TransformerEncoderLayer of models/sepformer.py:395-520) and its autograd backward, fp32 on the matrix pipe.
 * qkv, dqkv: (N, L, 3, H, D) -- the packed input projection's output as it is; o, dout: (N, L, H, D); lse, delta: (N, H, L) (lse written
 * forward, read backward; delta is scratch of the backward).  L <= 320, D in {8, 16, 32}.  p_drop: dropout rate on the probabilities
 * (0: none); the mask is a function of (seed, n, h, query, key), so forward and backward must be given the same seed. */
int sep_attn_fwd(const float* qkv, float* o, float* lse, int N, int L, int H, int D, float scale, float p_drop, unsigned long long seed,
                 sep_stream_t stream);
int sep_attn_bwd(const float* qkv, const float* o, const float* dout, const float* lse, float* delta, float* dqkv, int N, int L, int H, int D,
                 float scale, float p_drop, unsigned long long seed, sep_stream_t stream);

/* Layer norm over the features of token-major rows with the residual sum in front of it (ABI 22): nn.LayerNorm(C) as the post-norm
 * nn.TransformerEncoderLayer of SepFormer applies it (reference models/sepformer.py:395-520: `norm1(x + dropout1(self_attention(x)))`,
 * `norm2(x + dropout2(feed_forward(x)))`) and GALRNet's channel norm (models/galr.py:172-190 LayerNormAlongChannel), one pass each way.
 *   forward : s = x + drop(res) ; y = (s - mu) rstd gamma + beta with mu / biased variance of the row's C values, eps inside the root
 *   backward: ds = the gradient at s (it is the gradient at x AND at drop(res)) ; dres = the gradient at res
 * x, res, s, y, dy, ds, dres: (rows, C) fp32, features contiguous, C a multiple of 4 and <= 1024.  res NULL: y = LN(x), s must be NULL and the
 * backward is given x as s; s NULL with a branch: the sum is not kept (inference).  stat (rows, 2) = {mu, rstd}, written forward, read backward.  p_drop: rate of the inverted dropout on
 * res (0: none; then dres must be NULL -- ds serves both); the mask is a function of (seed, element index), the hash of sep_attn_*, so the
 * backward must be given the forward's seed.  part: (sep_rownorm_parts(rows, C), 2, C) = {sum dy * xhat | sum dy} over each workgroup's rows:
 * summed over the slabs they are d(gamma), d(beta). */
int sep_rownorm_parts(long rows, int C);
int sep_rownorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* s, float* y, float* stat, long rows, int C,
                    float eps, float p_drop, unsigned long long seed, sep_stream_t stream);
int sep_rownorm_bwd(const float* dy, const float* s, const float* gamma, const float* stat, float* ds, float* dres, float* part, long rows, int C,
                    float p_drop, unsigned long long seed, sep_stream_t stream);

/* ReLU + inverted dropout between the two Linear layers of the transformer layers' feed-forward sub-block (ABI 22; torch/nn/modules/
 * transformer.py _ff_block as models/sepformer.py:395-520 instantiates it): a = keep ? max(h, 0) / (1 - p) : 0 with the mask of sep_rownorm_*
 * (hash of (seed, element index)); backward dh = a != 0 ? dy / (1 - p) : 0 -- from the forward's OUTPUT, no mask, no seed.  n a multiple of 4. */
int sep_relu_drop_fwd(const float* h, float* a, long n, float p_drop, unsigned long long seed, sep_stream_t stream);
int sep_relu_drop_bwd(const float* dy, const float* a, float* dh, long n, float p_drop, sep_stream_t stream);

/* Stand-alone gLN (modules/norm.py:11-35) for callers outside the fused network. */
int sep_gln_stats(const float* x, double* stats, int B, int C, int T, int ldt, sep_stream_t stream);
int sep_gln_apply(const float* x, const double* stats, const float* gamma, const float* beta, float* y, int B, int C,
                  int T, int ldt, double count, float eps, sep_stream_t stream);
/* rowpart[b][c][ntile][2] = {sum_t dy, sum_t dy*x} per 1024-frame tile */
int sep_gln_bwd_rowsums(const float* dy, const float* x, float* rowpart, int B, int C, int T, int ldt,
                        sep_stream_t stream);
int sep_gln_bwd_apply(const float* dy, const float* x, const double* stats, const float* gamma, const float* bsum,
                      float* dx, int B, int C, int T, int ldt, double count, float eps, sep_stream_t stream);

/* Segment1d / OverlapAdd1d (models/transform.py:6-65) for rows = B*C padded rows of x (T valid frames, stride ldt):
 *   segment:     out[row][s][k] = xpad[row][s*hop + k]   (xpad = x shifted right by pad_left, zero elsewhere)
 *   overlap_add: out[row][t]    = sum_{s*hop + k == t + pad_left} y[row][s][k] for t < T, 0 for T <= t < ldt
 * adjoint pair: each one is the other's backward. */
int sep_segment(const float* x, float* out, int rows, int T, int ldt, int S, int chunk, int hop, int pad_left,
                sep_stream_t stream);
int sep_overlap_add(const float* y, float* out, int rows, int T, int ldt, int S, int chunk, int hop, int pad_left,
                    sep_stream_t stream);

/* Generic depthwise Conv1d of modules.conv.DepthwiseSeparableConv1d (src/modules/conv.py:13-29; nn.Conv1d(groups=C) with
 * kernel size Kw, stride, zero padding, dilation) on contiguous (B, C, T) tensors:
 *   y[b][c][to] = bias[c] + sum_k w[c][k] * xpad[b][c][to*stride + k*dil - pad]
 * bwd_weight writes partial[b][c][0..Kw-1] = dL/dw contributions and partial[b][c][Kw] = dL/dbias contribution of row (b,c). */
int sep_depthwise_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int Tin, int Tout, int Kw,
                      int stride, int pad, int dil, sep_stream_t stream);
int sep_depthwise_bwd_input(const float* dy, const float* w, float* dx, int B, int C, int Tin, int Tout, int Kw, int stride,
                            int pad, int dil, sep_stream_t stream);
int sep_depthwise_bwd_weight(const float* dy, const float* x, float* partial, int B, int C, int Tin, int Tout, int Kw,
                             int stride, int pad, int dil, sep_stream_t stream);

/* (B, C, T) <-> (B, C, ldt) repack with zero fill of the pad frames */
int sep_repack(const float* src, int ld_src, float* dst, int ld_dst, int rows, int T, sep_stream_t stream);

/* SI-SDR (criterion/sdr.py:122-139): per batch item and per (i, j) pair the three dot products
 *   dots[b][i][j] = <est_i, tgt_j>, tt[b][j] = |tgt_j|^2, xx[b][i] = |est_i|^2      (double)
 * all_pairs = 0 computes only i == j. */
int sep_sisdr_dots(const float* est, const float* tgt, double* dots, double* tt, double* xx, int B, int n, int T,
                   int all_pairs, sep_stream_t stream);
/* sisdr[b][i][j] (float) from the dot products, reference formula with eps. */
int sep_sisdr_from_dots(const double* dots, const double* tt, const double* xx, float* sisdr, int B, int n,
                        int all_pairs, float eps, sep_stream_t stream);
/* d_est[b][i][:] = sum_j gw[b][i][j] * d sisdr_ij / d est_i ; gw = dL/d sisdr (float, [B][n][n]; diagonal only if !all_pairs) */
int sep_sisdr_bwd(const float* est, const float* tgt, const double* dots, const double* tt, const double* xx,
                  const float* gw, float* d_est, int B, int n, int T, int all_pairs, float eps, sep_stream_t stream);

/* PIT search (criterion/pit.py:9-44) on the pair matrix: score[b][p] = reduce_s val[b][s][perm_p(s)] with
 * reduce = mean (or sum); picks min (maximize=0) or max; ties resolve to the first permutation in
 * itertools.permutations order like torch.min/max.  perms is [P][n] int32.  best_idx int64 [B], best_val float [B]. */
int sep_pit_search(const float* val, const int32_t* perms, int P, int n, int B, int maximize, int use_mean,
                   float* best_val, int64_t* best_idx, sep_stream_t stream);

/* Sinkhorn PIT (criterion/pit.py:163-193) on the cost matrix C[b][n][n] (float):
 * forward keeps every iterate in zwork (double [B][2*iters+1][n][n]) for the reverse sweep. */
int sep_sinkhorn_fwd(const float* C, double* zwork, float* loss, float* P, int B, int n, float coldness, int iters,
                     sep_stream_t stream);
int sep_sinkhorn_bwd(const float* C, const double* zwork, const float* dloss, float* dC, int B, int n, float coldness,
                     int iters, sep_stream_t stream);

/* The O(T) part of the waveform-distance criteria (reference src/criterion/distance.py:7-285: L1Loss, L2Loss,
 * SquaredError, MeanAbsoluteError, MeanSquaredError) and of plain SDR (src/criterion/sdr.py:6-22), over the last axis:
 *   sums[row] = { sum |x-t|, sum (x-t)^2, sum t^2 }   (double [rows][3]; x, t float [rows][T])
 *   dx[row][i] = c_abs[row] * sign(x-t) + c_sq[row] * (x-t)     (either coefficient vector may be NULL = 0)
 * the per-row value and coefficients are a few scalars per row formed by the caller exactly as the reference formula
 * states (mean / sqrt / log10 / eps placement).  rows < 65536 for the backward. */
int sep_rowdiff_sums(const float* x, const float* t, double* sums, int64_t rows, int T, sep_stream_t stream);
int sep_rowdiff_bwd(const float* x, const float* t, const float* c_abs, const float* c_sq, float* dx, int64_t rows, int T,
                    sep_stream_t stream);

/* clip_grad_norm_ + Adam of the train step (egs/wsj0-mix/common/src/driver.py:152-155), fused on a flat buffer:
 *   sqnorm[0] += sum g^2  (double, caller zeroes) ; then
 *   g *= min(1, max_norm/(sqrt(sqnorm)+1e-6)) (if max_norm > 0) ; Adam(lr, b1, b2, eps, weight_decay), step t. */
int sep_sqnorm(const float* g, double* sqnorm, int64_t n, sep_stream_t stream);
int sep_adam_step(float* p, float* g, float* m, float* v, const double* sqnorm, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float max_norm, float grad_scale, int step,
                  sep_stream_t stream);

/* The same step with its two per-step scalars in DEVICE memory, so that a captured launch (hipGraph of the whole train step)
 * replays correctly: step_dev[0] is incremented first (a one-thread kernel), then used for the bias corrections; lr_dev[0] is
 * read at run time (learning-rate halving of driver.py:101-111 is a host write into that word between replays). */
int sep_adam_step_dev(float* p, float* g, float* m, float* v, const double* sqnorm, int64_t n, const float* lr_dev,
                      int32_t* step_dev, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                      float grad_scale, sep_stream_t stream);

/* Time recurrence of ONE direction of nn.LSTM (reference src/models/dprnn.py:9-148 -> choose_rnn('lstm'), i.e.
 * IntraChunkRNN / InterChunkRNN of DPRNN-TasNet).  The input projection is done by the caller (a plain GEMM):
 *   xg[seq][t][4H] = x[seq][t][:] W_ih^T + b_ih + b_hh,   gate order i, f, g, o as in torch.
 * forward : a_t = xg_t + W_hh h_{t-1};  c_t = sigmoid(a_f) c_{t-1} + sigmoid(a_i) tanh(a_g);  h_t = sigmoid(a_o) tanh(c_t),
 *           h_0 = c_0 = 0, t ascending (reverse = 0) or descending (reverse = 1).  gates (post-activation i,f,g,o) and
 *           cstate are saved for the backward sweep; both may be NULL for inference.
 * backward: dh_out[seq][t][H] = gradient at h_t  ->  dxg[seq][t][4H] = gradient at the pre-activations a_t (= at xg_t).
 *           The caller forms dW_ih = dxg^T x, dW_hh = dxg^T h_{t-1}, db = sum dxg, dx = dxg W_ih (plain GEMMs).
 * H in {16, 32, 64, 128}; w_hh is [4H][H] row-major.
 * reverse = 2 runs BOTH directions in one launch (a sweep occupies only nseq/16 compute units): every buffer then holds
 * two slabs back to back, slab 0 = forward in time, slab 1 = backward in time (xg [2][nseq][L][4H], w_hh [2][4H][H], ...).
 * Two sweep kernels exist: sixteen sequences per workgroup (v_mfma_f32_16x16x4_f32) and four (v_mfma_f32_4x4x1_16b_f32); the
 * library picks four while those workgroups fit one round on the chip.  OR-ing SEP_LSTM_FORCE16 / SEP_LSTM_FORCE4 into
 * `reverse` forces one of them for the call (tests). */
#define SEP_LSTM_FORCE16 0x100
#define SEP_LSTM_FORCE4 0x200
/* with reverse = 2: h_out (sep_lstm_fwd) and dh_out (sep_lstm_bwd) are ONE (nseq, L, 2H) buffer, forward direction in columns [0, H),
 * reversed in [H, 2H) -- the layout nn.LSTM(bidirectional=True) returns -- instead of two (nseq, L, H) slabs: no torch.cat / torch.stack
 * around the sweeps.  gates, cstate, xg and dxg keep one slab per direction. */
#define SEP_LSTM_INTERLEAVED 0x400
int sep_lstm_fwd(const float* xg, const float* w_hh, float* h_out, float* gates, float* cstate, int nseq, int L, int H,
                 int reverse, sep_stream_t stream);
int sep_lstm_bwd(const float* dh_out, const float* gates, const float* cstate, const float* w_hh, float* dxg, int nseq,
                 int L, int H, int reverse, sep_stream_t stream);

/* Token-major dense layers of the dual-path separators (fp32 on the matrix pipe): the LSTMs' input projections and the nn.Linear behind
 * them at reference src/models/dprnn.py:65-148, with their gradients -- torch.addmm / `@` on [tokens][features] activations (features
 * contiguous) and torch's [N out][K in] weights.  K and N: multiples of 64 (sep_linear_fwd: K of 32; sep_linear_bwd_input: N of 32).
 *   sep_linear_fwd         y[t][n] = sum_k x[t][k] w[n][k] + bias[n] + bias2[n]          (bias, bias2 may be NULL)
 *   sep_linear_bwd_input   dx[t][k] = sum_n dy[t][n] w[n][k]                             (accumulate != 0: added to dx)
 *   sep_linear_bwd_weight  partial[s][n][k] = sum over the tokens of slab s of dy[t][n] x[t + shift][k],  s < nslab (the caller adds the
 *                          slabs: sep_reduce_slabs), partial_bias[s][n] = sum of dy[t][n] over the same tokens (may be NULL).
 *                          shift in {-1, 0, +1} and L: with shift != 0 the tokens are sequences of L steps (ntok % L == 0) and x[t + shift]
 *                          is the previous / next step's row of the SAME sequence, zero at its first / last step: the h_{t-1} operand of
 *                          the recurrent weights' gradient (dW_hh = sum_t dgates_t h_{t-1}^T) taken from h itself.  ldx: row stride of x in
 *                          floats (>= K, multiple of 4; x 16-byte aligned): one direction's half of an interleaved bi-LSTM output. */
int sep_linear_fwd(const float* x, const float* w, const float* bias, const float* bias2, float* y, long ntok, int K, int N,
                   sep_stream_t stream);
int sep_linear_bwd_input(const float* dy, const float* w, float* dx, long ntok, int K, int N, int accumulate, sep_stream_t stream);
int sep_linear_bwd_weight(const float* dy, const float* x, long ldx, float* partial, float* partial_bias, long ntok, int K, int N, int L,
                          int shift, int nslab, sep_stream_t stream);

/* The layout change in front of and behind those layers (reference src/models/dprnn.py:73-76,123-126, the permute / reshape pairs around
 * the intra- and inter-chunk recurrences), each the other's inverse and backward:
 *   sep_chunk_to_tokens   y[b][s][k][f] = x[b][f][s][k]   (inter = 0: sequences over k for every (b, s): y is (B*S, K, F))
 *                         y[b][k][s][f] = x[b][f][s][k]   (inter = 1: sequences over s for every (b, k): y is (B*K, S, F))
 *   sep_tokens_to_chunk   the reverse, x (B, F, S, K) contiguous. */
int sep_chunk_to_tokens(const float* x, float* y, int B, int F, int S, int K, int inter, sep_stream_t stream);
int sep_tokens_to_chunk(const float* y, float* x, int B, int F, int S, int K, int inter, sep_stream_t stream);

/* ---- recorded launch sequences (ABI 23) ------------------------------------------------------------------------------------
 * One C-ABI call per pass instead of one Python call per kernel.  The host runs a step ONCE through the ordinary entry points while it
 * records every call as a sep_seq_op -- the entry point's id (sep_seq_lookup) and its arguments in order, WITHOUT the trailing stream:
 * pointers (device buffers, and HOST descriptors / segment arrays the caller keeps alive and unchanged) in .p, integers in .i, float /
 * double parameters in .f.  sep_run_sequence then calls the same entry points again, in order, on `stream`; it stops at the first op that
 * fails and returns that op's code (sep_last_error names the op).  The buffers named by the ops must stay allocated at the recorded
 * addresses; values that change between runs live in device memory (sep_adam_step_dev's step count and learning rate, the input batch).
 * Replaces the per-launch Python of the train step of reference egs/wsj0-mix/common/src/driver.py:141-157 (model(mixture) ->
 * pit_criterion -> backward -> clip_grad_norm_ -> optimizer.step()); there is no FFI there to cite: the reference issues the same work as
 * ~900 ATen launches from the interpreter. */
#define SEP_SEQ_MAX_ARGS 26
typedef union sep_seq_arg {
    int64_t i;
    double f;
    const void* p;
} sep_seq_arg;
typedef struct sep_seq_op {
    int32_t fn;    /* sep_seq_lookup(name) */
    int32_t nargs; /* must equal sep_seq_nargs(fn) */
    sep_seq_arg args[SEP_SEQ_MAX_ARGS];
} sep_seq_op;
int sep_seq_count(void);              /* number of recordable entry points: ids are 0 .. count-1 */
int sep_seq_lookup(const char* name); /* id of the entry point `name`, -1 if it cannot be recorded (queries without a stream) */
const char* sep_seq_name(int fn);
int sep_seq_nargs(int fn); /* parameters of the entry point minus the stream */
int sep_run_sequence(const sep_seq_op* ops_host, int n, sep_stream_t stream);

/* What a fully recorded step needs where the eager step used torch kernels:
 *   sep_memset      hipMemsetAsync on the stream (statistics slots, fp64 accumulators, arrival counters: `torch.zeros` in the eager step)
 *   sep_absmax      out[0] = max |x[i]|: the A-operand bound a_amax of SEP_ARITH_F16X3 over the flat parameter buffer
 *   sep_pit_finish  tail of PIT (criterion/pit.py:33-44) behind sep_pit_search, and its backward: loss[0] = sign * mean_b best_val[b];
 *                   gw[b][i][j] = sign * scale if j == perms[best_idx[b]][i] else 0 (the dL/d sisdr matrix sep_sisdr_bwd takes: scale =
 *                   1 / (B n) for a mean over sources and batch); pattern[b][i] = perms[best_idx[b]][i] (int64).  loss, gw, pattern may be NULL. */
int sep_memset(void* dst, int value, size_t bytes, sep_stream_t stream);
int sep_absmax(const float* x, int64_t n, float* out, sep_stream_t stream);
int sep_pit_finish(const float* best_val, const int64_t* best_idx, const int32_t* perms, int P, int n, int B, float sign, float scale,
                   float* loss, float* gw, int64_t* pattern, sep_stream_t stream);
/*   sep_axpby       out[i] = a x[i] + b y[i] (y may be NULL; out may alias x or y): the sign flips between criterion kernels the eager step leaves to
 *                   torch -- SinkPIT's cost matrix C = -SI-SDR and dL / d SI-SDR = -dL / dC (criterion/pit.py:143-146).  With best_idx = perms = NULL
 *                   sep_pit_finish only forms loss[0] = sign * mean_b best_val[b] (the batch mean of SinkPIT's per-item losses, pit.py:155-156). */
int sep_axpby(const float* x, float a, const float* y, float b, float* out, int64_t n, sep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
