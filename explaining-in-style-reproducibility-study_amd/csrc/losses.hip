// losses.hip — the scalar loss reductions of the train step, each as ONE forward and ONE backward launch
// instead of the 5-10 ATen launches of its torch composition:
//   hinge            /root/reference/stylex/stylex_train.py:382-387   mean(relu(1 + real) + relu(1 - fake)),  fake.mean()
//   pl_lengths       :306-316 (:316)                                  sqrt(mean_l(sum_d(g^2)))  per sample
//   kl_logits        :421-438 + KLDivLoss(batchmean, log_target) :406 sum_b sum_k p_real (log p_real - log p_fake) / B
//   l1_mean          nn.L1Loss :404-405 (used at :415-418)            mean(|a - b|)
// All fp32 arithmetic; the L1 operands may be bf16 (the generated images of the bf16 mode).  Deterministic: fixed
// reduction trees, no atomics.  Scalars (the results and the incoming gradient of a backward) live in device memory —
// no host synchronisation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

namespace {

constexpr int LT = 256;  // threads per block of every kernel here

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the block's threads; the result is valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < LT / 64; ++w) r += sm[w];
    }
    __syncthreads();
    return r;
}

__device__ __forceinline__ float ld(const void* p, long i, int bf16) {
    if (bf16) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
    return reinterpret_cast<const float*>(p)[i];
}

__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even (finite values)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// ---- hinge ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LT) void hinge_fwd_kernel(const float* real, const float* fake, float* out, long n, int mode) {
    __shared__ float sm[LT / 64];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += LT) {
        if (mode == 0) s += fmaxf(1.f + real[i], 0.f) + fmaxf(1.f - fake[i], 0.f);
        else s += fake[i];
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

__global__ __launch_bounds__(LT) void hinge_bwd_kernel(const float* real, const float* fake, const float* gout, float* greal,
                                                        float* gfake, long n, int mode) {
    const long i = (long)blockIdx.x * LT + threadIdx.x;
    if (i >= n) return;
    const float g = gout[0] / (float)n;
    if (mode == 0) {  // relu'(0) = 0, as ATen's threshold_backward
        if (greal) greal[i] = 1.f + real[i] > 0.f ? g : 0.f;
        if (gfake) gfake[i] = 1.f - fake[i] > 0.f ? -g : 0.f;
    } else if (gfake) {
        gfake[i] = g;
    }
}

// ---- path lengths: one block per sample ------------------------------------------------------------------------------
__global__ __launch_bounds__(LT) void pl_lengths_fwd_kernel(const float* g, float* len, int L, int D) {
    __shared__ float sm[LT / 64];
    const long n = (long)L * D;
    const float* row = g + (long)blockIdx.x * n;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += LT) s += row[i] * row[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) len[blockIdx.x] = sqrtf(s / (float)L);
}

__global__ __launch_bounds__(LT) void pl_lengths_bwd_kernel(const float* g, const float* len, const float* glen, float* gg, int L,
                                                             int D) {
    const long n = (long)L * D;
    const long base = (long)blockIdx.x * n;
    // d sqrt(m) / d g = g / (L * sqrt(m)); a zero length gives 0 / 0 = NaN exactly as the torch composition does
    const float c = glen[blockIdx.x] / ((float)L * len[blockIdx.x]);
    for (long i = threadIdx.x; i < n; i += LT) gg[base + i] = c * g[base + i];
}

// ---- KL of two logit rows: one thread per sample, the batch reduced by one block ---------------------------------------
__device__ __forceinline__ void log_softmax_stats(const float* x, int K, float& mx, float& lse) {
    mx = x[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(x[k] - mx);
    lse = logf(s);
}

__device__ __forceinline__ float kl_row(const float* r, const float* f, int K) {
    float rm, rl, fm, fl;
    log_softmax_stats(r, K, rm, rl);
    log_softmax_stats(f, K, fm, fl);
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const float rlp = r[k] - rm - rl, flp = f[k] - fm - fl;
        s += expf(rlp) * (rlp - flp);
    }
    return s;
}

__global__ __launch_bounds__(LT) void kl_fwd_kernel(const float* real, const float* fake, float* out, int B, int K) {
    __shared__ float sm[LT / 64];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += LT) s += kl_row(real + (long)b * K, fake + (long)b * K, K);
    s = block_sum(s, sm);
    if (threadIdx.x == 0) out[0] = s / (float)B;
}

__global__ __launch_bounds__(LT) void kl_bwd_kernel(const float* real, const float* fake, const float* gout, float* greal, float* gfake,
                                                     int B, int K) {
    const int b = blockIdx.x * LT + threadIdx.x;
    if (b >= B) return;
    const float* r = real + (long)b * K;
    const float* f = fake + (long)b * K;
    float rm, rl, fm, fl;
    log_softmax_stats(r, K, rm, rl);
    log_softmax_stats(f, K, fm, fl);
    const float g = gout[0] / (float)B;
    float row = 0.f;
    if (greal) row = kl_row(r, f, K);
    for (int k = 0; k < K; ++k) {
        const float rlp = r[k] - rm - rl, flp = f[k] - fm - fl;
        const float p = expf(rlp), q = expf(flp);
        if (gfake) gfake[(long)b * K + k] = g * (q - p);                      // sum_k p_k = 1
        if (greal) greal[(long)b * K + k] = g * p * ((rlp - flp) - row);
    }
}

// ---- mean |a - b| over n elements: fixed grid of partial sums + one finishing block -------------------------------------
// The logical tensor is walked as a 4-D index {d0, d1, d2, d3} (d3 fastest); each operand is addressed either linearly
// in that order or through its own element strides (the real batch arrives NCHW, the generated one as the 3-channel
// slice of a 4-channel channels-last tensor).
constexpr int L1_BLOCKS = 1024;

struct L1Geo {
    int a_strided, b_strided;
    unsigned d1, d2, d3;  // extents of the three fastest dims of the walk
    long as[4], bs[4];    // element strides
};

__device__ __forceinline__ void l1_index(const L1Geo& g, long i, long& ia, long& ib) {
    ia = ib = i;
    if (!(g.a_strided | g.b_strided)) return;
    const unsigned u = (unsigned)i;  // n < 2^32 checked by the host
    const unsigned q3 = u / g.d3, i3 = u - q3 * g.d3;
    const unsigned q2 = q3 / g.d2, i2 = q3 - q2 * g.d2;
    const unsigned i0 = q2 / g.d1, i1 = q2 - i0 * g.d1;
    if (g.a_strided) ia = (long)i0 * g.as[0] + (long)i1 * g.as[1] + (long)i2 * g.as[2] + (long)i3 * g.as[3];
    if (g.b_strided) ib = (long)i0 * g.bs[0] + (long)i1 * g.bs[1] + (long)i2 * g.bs[2] + (long)i3 * g.bs[3];
}

__global__ __launch_bounds__(LT) void l1_partial_kernel(const void* a, const void* b, float* partial, long n, int abf, int bbf, L1Geo geo) {
    __shared__ float sm[LT / 64];
    float s = 0.f;
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) {
        long ia, ib;
        l1_index(geo, i, ia, ib);
        s += fabsf(ld(a, ia, abf) - ld(b, ib, bbf));
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(LT) void l1_finish_kernel(const float* partial, float* out, int nblocks, long n) {
    __shared__ float sm[LT / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += LT) s += partial[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

__global__ __launch_bounds__(LT) void l1_bwd_kernel(const void* a, const void* b, const float* gout, void* ga, void* gb, long n, int abf,
                                                     int bbf, L1Geo geo) {
    const float g = gout[0] / (float)n;
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) {
        long ia, ib;
        l1_index(geo, i, ia, ib);
        const float d = ld(a, ia, abf) - ld(b, ib, bbf);
        const float s = d > 0.f ? g : d < 0.f ? -g : 0.f;  // sign(0) = 0, as ATen
        if (ga) {
            if (abf) reinterpret_cast<unsigned short*>(ga)[ia] = f2bf(s);
            else reinterpret_cast<float*>(ga)[ia] = s;
        }
        if (gb) {
            if (bbf) reinterpret_cast<unsigned short*>(gb)[ib] = f2bf(-s);
            else reinterpret_cast<float*>(gb)[ib] = -s;
        }
    }
}

// shape4 NULL: both operands in one linear element order (then both stride arrays must be NULL too)
int l1_geo(const int64_t* shape4, const int64_t* a_strides4, const int64_t* b_strides4, int64_t n, L1Geo* g) {
    *g = L1Geo{0, 0, 1, 1, 1, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (!shape4) return (a_strides4 || b_strides4) ? STYLEX_EINVAL : 0;
    int64_t prod = 1;
    for (int k = 0; k < 4; ++k) {
        if (shape4[k] < 1 || (a_strides4 && a_strides4[k] < 0) || (b_strides4 && b_strides4[k] < 0)) return STYLEX_EINVAL;
        prod *= shape4[k];
    }
    if (prod != n || n >= (1ll << 32)) return STYLEX_EINVAL;
    g->a_strided = a_strides4 != nullptr;
    g->b_strided = b_strides4 != nullptr;
    g->d1 = (unsigned)shape4[1], g->d2 = (unsigned)shape4[2], g->d3 = (unsigned)shape4[3];
    for (int k = 0; k < 4; ++k) {
        g->as[k] = a_strides4 ? (long)a_strides4[k] : 0;
        g->bs[k] = b_strides4 ? (long)b_strides4[k] : 0;
    }
    return 0;
}

int l1_blocks(int64_t n) {
    int64_t b = (n + LT * 8 - 1) / (LT * 8);
    return (int)(b < 1 ? 1 : b > L1_BLOCKS ? L1_BLOCKS : b);
}

}  // namespace

extern "C" {

int stylex_hinge_fwd(const float* real, const float* fake, float* out, int64_t n, int mode, void* stream) {
    if (!fake || !out || n < 1 || (mode != 0 && mode != 1) || (mode == 0 && !real)) return STYLEX_EINVAL;
    stylex_note_kernel("hinge_fwd_kernel");
    hipLaunchKernelGGL(hinge_fwd_kernel, dim3(1), dim3(LT), 0, (hipStream_t)stream, real, fake, out, (long)n, mode);
    return (int)hipGetLastError();
}

int stylex_hinge_bwd(const float* real, const float* fake, const float* gout, float* greal, float* gfake, int64_t n, int mode,
                     void* stream) {
    if (!fake || !gout || n < 1 || (mode != 0 && mode != 1) || (mode == 0 && !real) || (!greal && !gfake)) return STYLEX_EINVAL;
    stylex_note_kernel("hinge_bwd_kernel");
    hipLaunchKernelGGL(hinge_bwd_kernel, dim3((unsigned)((n + LT - 1) / LT)), dim3(LT), 0, (hipStream_t)stream, real, fake, gout, greal, gfake, (long)n, mode);
    return (int)hipGetLastError();
}

// shape = [B, L, D]
int stylex_pl_lengths_fwd(const float* g, float* len, const int64_t* shape, void* stream) {
    if (!g || !len || !shape || shape[0] < 1 || shape[1] < 1 || shape[2] < 1 || shape[1] * shape[2] > (1ll << 30)) return STYLEX_EINVAL;
    stylex_note_kernel("pl_lengths_fwd_kernel");
    hipLaunchKernelGGL(pl_lengths_fwd_kernel, dim3((unsigned)shape[0]), dim3(LT), 0, (hipStream_t)stream, g, len, (int)shape[1], (int)shape[2]);
    return (int)hipGetLastError();
}

int stylex_pl_lengths_bwd(const float* g, const float* len, const float* glen, float* gg, const int64_t* shape, void* stream) {
    if (!g || !len || !glen || !gg || !shape || shape[0] < 1 || shape[1] < 1 || shape[2] < 1 || shape[1] * shape[2] > (1ll << 30))
        return STYLEX_EINVAL;
    stylex_note_kernel("pl_lengths_bwd_kernel");
    hipLaunchKernelGGL(pl_lengths_bwd_kernel, dim3((unsigned)shape[0]), dim3(LT), 0, (hipStream_t)stream, g, len, glen, gg, (int)shape[1], (int)shape[2]);
    return (int)hipGetLastError();
}

// shape = [B, K]
int stylex_kl_logits_fwd(const float* real, const float* fake, float* out, const int64_t* shape, void* stream) {
    if (!real || !fake || !out || !shape || shape[0] < 1 || shape[1] < 1 || shape[0] > (1 << 24) || shape[1] > (1 << 16)) return STYLEX_EINVAL;
    stylex_note_kernel("kl_fwd_kernel");
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(1), dim3(LT), 0, (hipStream_t)stream, real, fake, out, (int)shape[0], (int)shape[1]);
    return (int)hipGetLastError();
}

int stylex_kl_logits_bwd(const float* real, const float* fake, const float* gout, float* greal, float* gfake, const int64_t* shape,
                         void* stream) {
    if (!real || !fake || !gout || (!greal && !gfake) || !shape || shape[0] < 1 || shape[1] < 1 || shape[0] > (1 << 24) ||
        shape[1] > (1 << 16))
        return STYLEX_EINVAL;
    stylex_note_kernel("kl_bwd_kernel");
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)((shape[0] + LT - 1) / LT)), dim3(LT), 0, (hipStream_t)stream, real, fake, gout, greal, gfake,
                       (int)shape[0], (int)shape[1]);
    return (int)hipGetLastError();
}

// floats of workspace stylex_l1_mean_fwd needs for n elements
int64_t stylex_l1_mean_chunks(int64_t n) { return n < 1 ? 0 : l1_blocks(n); }

// a_dtype / b_dtype: 0 = fp32, 1 = bf16.  shape4 = NULL: a and b (and their gradients) share one linear element order.
// Otherwise the logical tensor is the 4-D index space {shape4} (n = its product < 2^32), walked with the last index
// fastest; an operand whose stride array is NULL is linear in that order, the other is addressed through its element
// strides (its gradient is written at the same offsets).
int stylex_l1_mean_fwd(const void* a, const void* b, float* partial, float* out, int64_t n, int a_dtype, int b_dtype,
                       const int64_t* shape4, const int64_t* a_strides4, const int64_t* b_strides4, void* stream) {
    if (!a || !b || !partial || !out || n < 1 || (a_dtype | b_dtype) & ~1) return STYLEX_EINVAL;
    L1Geo geo;
    if (int rc = l1_geo(shape4, a_strides4, b_strides4, n, &geo)) return rc;
    const int nb = l1_blocks(n);
    stylex_note_kernel("l1_partial_kernel");
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(LT), 0, (hipStream_t)stream, a, b, partial, (long)n, a_dtype, b_dtype, geo);
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(LT), 0, (hipStream_t)stream, (const float*)partial, out, nb, (long)n);
    return (int)hipGetLastError();
}

// ga / gb (either may be null) take the dtype and the addressing of their operand
int stylex_l1_mean_bwd(const void* a, const void* b, const float* gout, void* ga, void* gb, int64_t n, int a_dtype, int b_dtype,
                       const int64_t* shape4, const int64_t* a_strides4, const int64_t* b_strides4, void* stream) {
    if (!a || !b || !gout || (!ga && !gb) || n < 1 || (a_dtype | b_dtype) & ~1) return STYLEX_EINVAL;
    L1Geo geo;
    if (int rc = l1_geo(shape4, a_strides4, b_strides4, n, &geo)) return rc;
    stylex_note_kernel("l1_bwd_kernel");
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(l1_blocks(n)), dim3(LT), 0, (hipStream_t)stream, a, b, gout, ga, gb, (long)n, a_dtype, b_dtype, geo);
    return (int)hipGetLastError();
}

}  // extern "C"
