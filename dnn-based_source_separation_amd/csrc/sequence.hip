// Recorded launch sequences (ABI 23): one C-ABI call runs a whole pass of the Conv-TasNet step.
//
// The step of reference egs/wsj0-mix/common/src/driver.py:141-157 (forward, criterion, backward, clip, Adam) is ~360 kernel launches of
// fixed shapes; driven from Python every launch costs ~30 us of interpreter time (descriptor marshalling, allocation, argument checks),
// which bounds the step below ~10 utterances per GPU (the recipes train with 2 - 4: train.sh:54).  The host records the entry points it
// calls ONCE per shape -- (function id, arguments) in `sep_seq_op` records, descriptors kept alive on the host side -- and afterwards
// hands the list to sep_run_sequence, which calls the same entry points again from a C loop on the stream it is given.  Nothing is
// captured by the runtime (no hipGraph: its replays of the full-size step were measured wrong on this stack,
// profiles/r07_round5_experiments.md r07m); every replayed op is an ordinary launch through the library's own entry point, with the
// entry point's own argument checks.
//
// Also here: the three small entry points a fully recorded step needs where the eager step used torch kernels -- sep_memset (zeroing
// of statistics / accumulators), sep_absmax (the A-operand bound of SEP_ARITH_F16X3 over the flat parameter buffer) and
// sep_pit_finish (batch mean of the PIT search's best values and the gradient weights of the chosen permutation: the tail of
// reference src/criterion/pit.py:33-44 and its backward).
#include "common.hpp"
#include <string.h>
#include <type_traits>
#include <utility>

namespace {

typedef int (*seq_thunk)(const sep_seq_arg*, sep_stream_t);

template <class T, bool IS_STREAM>
inline T seq_pick(const sep_seq_arg* a, const size_t i, sep_stream_t s) {
    if constexpr (IS_STREAM) {
        return (T)s;
    } else if constexpr (std::is_pointer<T>::value) {
        return (T)a[i].p;
    } else if constexpr (std::is_floating_point<T>::value) {
        return (T)a[i].f;
    } else {
        return (T)a[i].i;
    }
}
template <class... A, size_t... I>
inline int seq_invoke(int (*fn)(A...), const sep_seq_arg* a, sep_stream_t s, std::index_sequence<I...>) {
    return fn(seq_pick<A, I + 1 == sizeof...(A)>(a, I, s)...);      // the trailing parameter of every recordable entry point is the stream
}
template <class... A>
constexpr int seq_arity(int (*)(A...)) { return (int)sizeof...(A) - 1; }

struct seq_entry { const char* name; seq_thunk call; int nargs; };

#define SEQ_FN(f)                                                                                                                          \
    { #f, [](const sep_seq_arg* a, sep_stream_t s) -> int { return seq_invoke(&f, a, s, std::make_index_sequence<seq_arity(&f) + 1>{}); }, \
      seq_arity(&f) }

// every entry point of include/sepkernels.h whose last parameter is the stream (the queries without one are not launches)
const seq_entry SEQ_TABLE[] = {
    SEQ_FN(sep_pw_gemm), SEQ_FN(sep_pack_weights), SEQ_FN(sep_pw_wgrad), SEQ_FN(sep_pw_wgrad_batch), SEQ_FN(sep_reduce_slabs), SEQ_FN(sep_f64_to_f32),
    SEQ_FN(sep_encoder_fwd), SEQ_FN(sep_unfold), SEQ_FN(sep_dwconv_fwd), SEQ_FN(sep_dwconv_bwd), SEQ_FN(sep_gln_bwd_finalize),
    SEQ_FN(sep_gln_bwd_finalize_batch), SEQ_FN(sep_gln_bwd_from_wgrad), SEQ_FN(sep_head_bwd), SEQ_FN(sep_decoder_fwd),
    SEQ_FN(sep_decoder_bwd), SEQ_FN(sep_softmax_ch_fwd), SEQ_FN(sep_softmax_ch_bwd), SEQ_FN(sep_cln_fwd), SEQ_FN(sep_cln_bwd),
    SEQ_FN(sep_gln_tokens_fwd), SEQ_FN(sep_gln_tokens_bwd), SEQ_FN(sep_attn_fwd), SEQ_FN(sep_attn_bwd), SEQ_FN(sep_rownorm_fwd),
    SEQ_FN(sep_rownorm_bwd), SEQ_FN(sep_relu_drop_fwd), SEQ_FN(sep_relu_drop_bwd), SEQ_FN(sep_gln_stats), SEQ_FN(sep_gln_apply),
    SEQ_FN(sep_gln_bwd_rowsums), SEQ_FN(sep_gln_bwd_apply), SEQ_FN(sep_segment), SEQ_FN(sep_overlap_add), SEQ_FN(sep_depthwise_fwd),
    SEQ_FN(sep_depthwise_bwd_input), SEQ_FN(sep_depthwise_bwd_weight), SEQ_FN(sep_repack), SEQ_FN(sep_sisdr_dots),
    SEQ_FN(sep_sisdr_from_dots), SEQ_FN(sep_sisdr_bwd), SEQ_FN(sep_pit_search), SEQ_FN(sep_sinkhorn_fwd), SEQ_FN(sep_sinkhorn_bwd),
    SEQ_FN(sep_rowdiff_sums), SEQ_FN(sep_rowdiff_bwd), SEQ_FN(sep_sqnorm), SEQ_FN(sep_adam_step), SEQ_FN(sep_adam_step_dev),
    SEQ_FN(sep_lstm_fwd), SEQ_FN(sep_lstm_bwd), SEQ_FN(sep_linear_fwd), SEQ_FN(sep_linear_bwd_input), SEQ_FN(sep_linear_bwd_weight),
    SEQ_FN(sep_chunk_to_tokens), SEQ_FN(sep_tokens_to_chunk), SEQ_FN(sep_memset), SEQ_FN(sep_absmax), SEQ_FN(sep_pit_finish), SEQ_FN(sep_axpby), SEQ_FN(sep_split_rows),
};
constexpr int SEQ_COUNT = (int)(sizeof(SEQ_TABLE) / sizeof(SEQ_TABLE[0]));

// max |x| over n floats: per-workgroup maxima, the non-negative floats compared as integers by one atomicMax each (out zeroed first)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, const int64_t n, unsigned* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = fabsf(x[i]);
        m = v > m ? v : m;                          // (a NaN never wins: the bound then comes from the finite values, as torch's amax would not --
    }                                               //  but a NaN weight has already ruined the step)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// One workgroup: loss = sign * mean_b best_val[b] (fixed summation order); gw[b][i][j] = sign * scale where j = perms[best_idx[b]][i], else 0
__global__ __launch_bounds__(256) void pit_finish_kernel(const float* __restrict__ best_val, const int64_t* __restrict__ best_idx,
                                                         const int32_t* __restrict__ perms, const int B, const int n, const float sign,
                                                         const float scale, float* __restrict__ loss, float* __restrict__ gw,
                                                         int64_t* __restrict__ pattern) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) acc += (double)best_val[b];
    const double tot = block_sum_256<double>(acc, red);
    if (threadIdx.x == 0 && loss) loss[0] = (float)((double)sign * tot / (double)B);
    if (!gw && !pattern) return;
    for (int e = threadIdx.x; e < B * n * n; e += 256) {
        const int b = e / (n * n), i = (e / n) % n, j = e % n;
        const int sel = perms[(size_t)best_idx[b] * n + i];
        if (gw) gw[e] = j == sel ? sign * scale : 0.f;
        if (pattern && j == 0) pattern[(size_t)b * n + i] = sel;
    }
}

// out[i] = a * x[i] + b * y[i] (y may be NULL): the sign flips and sums between criterion kernels that the eager step leaves to torch
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float a, const float* __restrict__ y, const float b,
                                                    float* __restrict__ out, const int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = y ? a * x[i] + b * y[i] : a * x[i];
}

}  // namespace

extern "C" int sep_seq_count(void) { return SEQ_COUNT; }

extern "C" int sep_seq_lookup(const char* name) {
    if (name)
        for (int k = 0; k < SEQ_COUNT; ++k)
            if (strcmp(SEQ_TABLE[k].name, name) == 0) return k;
    return -1;
}

extern "C" const char* sep_seq_name(int fn) { return fn >= 0 && fn < SEQ_COUNT ? SEQ_TABLE[fn].name : nullptr; }

extern "C" int sep_seq_nargs(int fn) { return fn >= 0 && fn < SEQ_COUNT ? SEQ_TABLE[fn].nargs : -1; }

extern "C" int sep_run_sequence(const sep_seq_op* ops_host, int n, sep_stream_t stream) {
    SEP_REQUIRE(n >= 0 && (ops_host || n == 0), "sep_run_sequence: bad arguments");
    for (int k = 0; k < n; ++k) {
        const sep_seq_op& op = ops_host[k];
        SEP_REQUIRE(op.fn >= 0 && op.fn < SEQ_COUNT, "sep_run_sequence: op %d names entry point %d (of %d)", k, op.fn, SEQ_COUNT);
        SEP_REQUIRE(op.nargs == SEQ_TABLE[op.fn].nargs, "sep_run_sequence: op %d (%s) carries %d arguments, the entry point takes %d", k,
                    SEQ_TABLE[op.fn].name, op.nargs, SEQ_TABLE[op.fn].nargs);
        const int rc = SEQ_TABLE[op.fn].call(op.args, stream);
        if (rc != 0) {
            char inner[400];
            strncpy(inner, sep_last_error(), sizeof(inner) - 1);
            inner[sizeof(inner) - 1] = 0;
            sep_set_error("sep_run_sequence: op %d of %d (%s) failed (%d): %s", k, n, SEQ_TABLE[op.fn].name, rc, inner);
            return rc;
        }
    }
    return 0;
}

extern "C" int sep_memset(void* dst, int value, size_t bytes, sep_stream_t stream) {
    SEP_REQUIRE(dst || bytes == 0, "sep_memset: bad arguments");
    if (bytes == 0) return 0;
    SEP_REQUIRE(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream) == hipSuccess, "sep_memset: hipMemsetAsync failed");
    return 0;
}

extern "C" int sep_absmax(const float* x, int64_t n, float* out, sep_stream_t stream) {
    SEP_REQUIRE(x && out && n > 0, "sep_absmax: bad arguments");
    SEP_REQUIRE(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream) == hipSuccess, "sep_absmax: clearing the result failed");
    const int64_t want = (n + 256 * 8 - 1) / (256 * 8);
    const int grid = (int)(want < 1 ? 1 : want > 1024 ? 1024 : want);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, reinterpret_cast<unsigned*>(out));
    SEP_CHECK_LAUNCH("sep_absmax");
    return 0;
}

extern "C" int sep_pit_finish(const float* best_val, const int64_t* best_idx, const int32_t* perms, int P, int n, int B, float sign,
                              float scale, float* loss, float* gw, int64_t* pattern, sep_stream_t stream) {
    SEP_REQUIRE(best_val && n > 0 && B > 0 && (loss || gw || pattern), "sep_pit_finish: bad arguments");
    SEP_REQUIRE((best_idx && perms && P > 0) || (!gw && !pattern), "sep_pit_finish: gw / pattern need best_idx and perms");
    hipLaunchKernelGGL(pit_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, best_val, best_idx, perms, B, n, sign, scale, loss, gw, pattern);
    SEP_CHECK_LAUNCH("sep_pit_finish");
    return 0;
}

extern "C" int sep_axpby(const float* x, float a, const float* y, float b, float* out, int64_t n, sep_stream_t stream) {
    SEP_REQUIRE(x && out && n > 0, "sep_axpby: bad arguments");
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)(want > 2048 ? 2048 : want)), dim3(256), 0, (hipStream_t)stream, x, a, y, b, out, n);
    SEP_CHECK_LAUNCH("sep_axpby");
    return 0;
}
