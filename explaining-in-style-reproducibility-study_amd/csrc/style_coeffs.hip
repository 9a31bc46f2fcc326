// style_coeffs.hip — the per-sample coefficients of a modulated convolution (Conv2DMod.forward,
// /root/reference/stylex/stylex_train.py:650-656 in the batched form ops.mod_coeffs documents):
//     s1[b][i] = style[b][i] + 1
//     d[b][o]  = rsqrt( sum_i s1[b][i]^2 * wsq[o][i] + eps ),   wsq[o][i] = sum_k W[o][i][k]^2
// and their first-order backward.  SURVEY §8(b) names `demod_coeff` / `bwd_style` as kernels of the set; as ATen ops they
// were ~8 launches forward and ~15 backward per layer call (x 28 calls per step: ~650 tiny launches).  Here: one
// launch forward (plus one `wsq` launch per weight VERSION), two backward.  fp32 throughout, fixed summation order.
//
// These are small dense contractions ([B<=128] x [C<=512] x [O<=512]) that run next to the conv chain: the kernels are
// sized for latency (many small blocks, coalesced reads of the L2-resident wsq), not for FLOPs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

namespace {

__device__ __forceinline__ float wave_sum16(float v) {  // sum over the 16 lanes of a row group
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
}

// wsq[oi] = sum_k w[oi][k]^2
__global__ __launch_bounds__(256) void wsq_kernel(const float* __restrict__ w, float* __restrict__ wsq, int OI, int K) {
    const int oi = blockIdx.x * 256 + threadIdx.x;
    if (oi >= OI) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const float v = w[(long)oi * K + k];
        s = fmaf(v, v, s);
    }
    wsq[oi] = s;
}

// grid (ceil(O / 16), B), block 256 = 16 output channels x 16 lanes over the input channels
__global__ __launch_bounds__(256) void modcoeff_fwd_kernel(const float* __restrict__ style, const float* __restrict__ wsq,
                                                           float* __restrict__ s1, float* __restrict__ d, int C, int O, float eps) {
    const int b = blockIdx.y, lane = threadIdx.x & 15, orow = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + orow;
    const float* st = style + (long)b * C;
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < C; i += 256) s1[(long)b * C + i] = st[i] + 1.f;
    float acc = 0.f;
    if (o < O) {
        const float* wr = wsq + (long)o * C;
        for (int i = lane; i < C; i += 16) {
            const float s = st[i] + 1.f;
            acc = fmaf(s * s, wr[i], acc);
        }
    }
    acc = wave_sum16(acc);
    if (lane == 0 && o < O) d[(long)b * O + o] = rsqrtf(acc + eps);
}

// gs[b][i] = (gs1 ? gs1[b][i] : 0) + 2 s1[b][i] * sum_o dq[b][o] wsq[o][i],   dq = -0.5 gd d^3
// grid (ceil(C / 64), B), block 256 = 4 groups over o x 64 input channels (coalesced rows of wsq)
__global__ __launch_bounds__(256) void modcoeff_bwd_style_kernel(const float* __restrict__ gd, const float* __restrict__ d,
                                                                 const float* __restrict__ s1, const float* __restrict__ wsq,
                                                                 const float* __restrict__ gs1, float* __restrict__ gs, int C,
                                                                 int O) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, il = threadIdx.x & 63, og = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float acc = 0.f;
    if (i < C)
        for (int o = og; o < O; o += 4) {
            const float dv = d[(long)b * O + o];
            const float dq = -0.5f * gd[(long)b * O + o] * dv * dv * dv;
            acc = fmaf(dq, wsq[(long)o * C + i], acc);
        }
    part[og][il] = acc;
    __syncthreads();
    if (og == 0 && i < C) {
        const float t = (part[0][il] + part[1][il]) + (part[2][il] + part[3][il]);
        const long k = (long)b * C + i;
        gs[k] = (gs1 ? gs1[k] : 0.f) + 2.f * s1[k] * t;
    }
}

// gw[o][i][k] = 2 w[o][i][k] * sum_b dq[b][o] s1[b][i]^2
// grid (ceil(C / 64), ceil(O / 4)), block 256 = 4 output channels x 64 input channels
__global__ __launch_bounds__(256) void modcoeff_bwd_weight_kernel(const float* __restrict__ gd, const float* __restrict__ d,
                                                                  const float* __restrict__ s1, const float* __restrict__ w,
                                                                  float* __restrict__ gw, int B, int C, int O, int K) {
    const int il = threadIdx.x & 63, ol = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il, o = blockIdx.y * 4 + ol;
    if (i >= C || o >= O) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float dv = d[(long)b * O + o];
        const float dq = -0.5f * gd[(long)b * O + o] * dv * dv * dv;
        const float s = s1[(long)b * C + i];
        acc = fmaf(dq, s * s, acc);
    }
    const long base = ((long)o * C + i) * K;
    for (int k = 0; k < K; ++k) gw[base + k] = 2.f * w[base + k] * acc;
}

}  // namespace

extern "C" {

int stylex_weight_sumsq(const float* w, float* wsq, int64_t O, int64_t C, int64_t K, void* stream) {
    if (!w || !wsq || O < 1 || C < 1 || K < 1 || O * C > 0x7fffffff) return STYLEX_EINVAL;
    const int OI = (int)(O * C);
    hipLaunchKernelGGL(wsq_kernel, dim3((OI + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wsq, OI, (int)K);
    return (int)hipGetLastError();
}

int stylex_modcoeff_fwd(const float* style, const float* wsq, float* s1, float* d, int64_t B, int64_t C, int64_t O, float eps,
                        void* stream) {
    if (!style || !wsq || !s1 || !d || B < 1 || C < 1 || O < 1 || B > 65535) return STYLEX_EINVAL;
    hipLaunchKernelGGL(modcoeff_fwd_kernel, dim3((unsigned)((O + 15) / 16), (unsigned)B), dim3(256), 0, (hipStream_t)stream, style,
                       wsq, s1, d, (int)C, (int)O, eps);
    return (int)hipGetLastError();
}

int stylex_modcoeff_bwd(const float* gd, const float* d, const float* s1, const float* wsq, const float* w, const float* gs1,
                        float* gstyle, float* gw, int64_t B, int64_t C, int64_t O, int64_t K, void* stream) {
    if (!gd || !d || !s1 || !wsq || B < 1 || C < 1 || O < 1 || K < 1 || B > 65535) return STYLEX_EINVAL;
    if (gstyle)
        hipLaunchKernelGGL(modcoeff_bwd_style_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, gd, d, s1, wsq, gs1, gstyle, (int)C, (int)O);
    if (gw) {
        if (!w) return STYLEX_EINVAL;
        hipLaunchKernelGGL(modcoeff_bwd_weight_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((O + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, gd, d, s1, w, gw, (int)B, (int)C, (int)O, (int)K);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
