// adam_pack.hip — multi-tensor Adam step that also rewrites the packed operand copies of the weights it updates.
//
// Reference: StylEx.G_opt / D_opt = Adam(lr, betas=(0.5, 0.9)) (/root/reference/stylex/stylex_train.py:957-959), stepped
// once per phase (:1357, :1449).  In the bf16 speed mode every conv weight has derived copies that the kernels read
// instead of the fp32 master: the K-contiguous bf16 operand layouts [N][tap][C] / [C][tap][N] (stylex_pack_weight), their
// space-to-depth form for the stride-2 convs (stylex_pack_weight_s2d), the bf16 [N][C] matrix of the 1x1 residual GEMMs,
// and the per-(o, i) sum of squares behind the demodulation coefficient (stylex_weight_sumsq).  Until round 3 those were
// rebuilt lazily after every optimiser step: ~90 launches of 12-25 us per train() call on the critical chain (1.4 ms),
// after a fused Adam that had just had every weight in registers.  Here ONE launch per optimiser step applies Adam
// (the update rule of torch._fused_adam_: bias corrections in double, eps outside the square root)
// and writes every registered derived copy from the freshly updated value.
//
// Work decomposition: a conv weight [N][C][T] is cut into slabs of 8 output channels x 64 input channels x T taps (one
// block each): the fp32 reads / writes of p, g, m, v are contiguous runs of 64 T floats per output channel, the updated
// values are parked in LDS (18 KB), and each layout is written with the index that is contiguous IN THAT LAYOUT fastest
// (128-byte runs of the forward layout, 16-byte runs of the transposed one).  Flat tensors (biases, linears) take 4608
// elements per block.  HBM-bound: 28 B per parameter + 2-8 B of copies.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

namespace {

constexpr int SLAB_N = 8, SLAB_C = 64, MAX_T = 9, SLAB_ELEMS = SLAB_N * SLAB_C * MAX_T;  // 4608

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

struct AdamConsts {
    float step_size, bc2_sqrt, beta1, beta2, eps;
};

__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, const AdamConsts& k) {
    // torch/ATen fused_adam_utils.cuh (ADAM_MODE::ORIGINAL, no amsgrad / weight decay / maximize):
    //   exp_avg = beta1 * exp_avg + (1 - beta1) * grad;  exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad
    //   denom = sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps;  param -= lr / bias_correction1 * exp_avg / denom
    // Explicitly rounded operations: with the compiler free to contract a * b + c into an FMA, the 16-byte and the
    // scalar code paths of the kernel rounded differently in the last place — and a data-parallel run (gradients are
    // 4-byte-aligned views of the flat buckets) stopped being bit-identical to a single-GPU run (separately allocated
    // gradients), which tests/test_hip_parity.py::test_gradsync_on_one_rank_rccl_is_bit_identical_to_single_gpu caught.
    m = __fmaf_rn(k.beta1, m, __fmul_rn(1.f - k.beta1, g));
    v = __fmaf_rn(k.beta2, v, __fmul_rn(__fmul_rn(1.f - k.beta2, g), g));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), k.bc2_sqrt), k.eps);
    return __fsub_rn(p, __fdiv_rn(__fmul_rn(k.step_size, m), denom));
}

__global__ __launch_bounds__(256) void adam_pack_kernel(const stylex_adam_tensor* __restrict__ descs,
                                                        const int32_t* __restrict__ block_map) {
    __shared__ float sw[SLAB_ELEMS];
    // by reference: a by-value copy of the 200-byte descriptor, indexed by the variant loop, lived in scratch memory
    const stylex_adam_tensor& d = descs[block_map[blockIdx.x]];
    const int local = (int)((int64_t)blockIdx.x - d.first_block);
    const int tid = threadIdx.x;
    AdamConsts k;
    {
        const double step = (double)d.step[0];
        const double bc1 = 1.0 - pow(d.beta1, step), bc2 = 1.0 - pow(d.beta2, step);
        k.step_size = (float)(d.lr / bc1);
        k.bc2_sqrt = (float)sqrt(bc2);
        k.beta1 = (float)d.beta1;
        k.beta2 = (float)d.beta2;
        k.eps = (float)d.eps;
    }
    if (d.N == 0) {  // flat tensor
        const int64_t e0 = (int64_t)local * SLAB_ELEMS;
        if (e0 + SLAB_ELEMS <= d.numel && (((uintptr_t)d.p | (uintptr_t)d.g | (uintptr_t)d.m | (uintptr_t)d.v) & 15) == 0) {
            for (int e = tid * 4; e < SLAB_ELEMS; e += 1024) {
                const int64_t i = e0 + e;
                float4 p4 = *reinterpret_cast<const float4*>(d.p + i), g4 = *reinterpret_cast<const float4*>(d.g + i);
                float4 m4 = *reinterpret_cast<const float4*>(d.m + i), v4 = *reinterpret_cast<const float4*>(d.v + i);
                p4.x = adam_update(p4.x, g4.x, m4.x, v4.x, k);
                p4.y = adam_update(p4.y, g4.y, m4.y, v4.y, k);
                p4.z = adam_update(p4.z, g4.z, m4.z, v4.z, k);
                p4.w = adam_update(p4.w, g4.w, m4.w, v4.w, k);
                *reinterpret_cast<float4*>(d.p + i) = p4;
                *reinterpret_cast<float4*>(d.m + i) = m4;
                *reinterpret_cast<float4*>(d.v + i) = v4;
            }
            return;
        }
        for (int e = tid; e < SLAB_ELEMS; e += 256) {
            const int64_t i = e0 + e;
            if (i >= d.numel) break;
            float m = d.m[i], v = d.v[i];
            d.p[i] = adam_update(d.p[i], d.g[i], m, v, k);
            d.m[i] = m;
            d.v[i] = v;
        }
        return;
    }
    const int N = d.N, C = d.C, T = d.T;
    const int ncs = (C + SLAB_C - 1) / SLAB_C;
    const int n0 = (local / ncs) * SLAB_N, c0 = (local % ncs) * SLAB_C;
    const int nn = min(SLAB_N, N - n0), cw = min(SLAB_C, C - c0);
    const int row = cw * T;  // contiguous run of one output channel inside the slab
    // 16-byte accesses (every slab of the step's conv weights) — only on 16-byte aligned tensors: under data parallelism
    // the gradient is a view into a flat bucket (parallel.py aligns its views since round 5; any other caller's need not be)
    const bool aligned16 = ((((uintptr_t)d.p | (uintptr_t)d.g | (uintptr_t)d.m | (uintptr_t)d.v) & 15) == 0);
    if (aligned16 && (row & 3) == 0 && ((C * T) & 3) == 0) {
        const int row4 = row >> 2;
        for (int e = tid; e < nn * row4; e += 256) {
            const int nl = e / row4, r = (e - nl * row4) << 2;
            const int64_t i = ((int64_t)(n0 + nl) * C + c0) * T + r;
            float4 p4 = *reinterpret_cast<const float4*>(d.p + i), g4 = *reinterpret_cast<const float4*>(d.g + i);
            float4 m4 = *reinterpret_cast<const float4*>(d.m + i), v4 = *reinterpret_cast<const float4*>(d.v + i);
            p4.x = adam_update(p4.x, g4.x, m4.x, v4.x, k);
            p4.y = adam_update(p4.y, g4.y, m4.y, v4.y, k);
            p4.z = adam_update(p4.z, g4.z, m4.z, v4.z, k);
            p4.w = adam_update(p4.w, g4.w, m4.w, v4.w, k);
            *reinterpret_cast<float4*>(d.p + i) = p4;
            *reinterpret_cast<float4*>(d.m + i) = m4;
            *reinterpret_cast<float4*>(d.v + i) = v4;
            *reinterpret_cast<float4*>(sw + nl * (SLAB_C * MAX_T) + r) = p4;
        }
    } else {
        for (int e = tid; e < nn * row; e += 256) {
            const int nl = e / row, r = e - nl * row;
            const int64_t i = ((int64_t)(n0 + nl) * C + c0) * T + r;
            float m = d.m[i], v = d.v[i];
            const float w = adam_update(d.p[i], d.g[i], m, v, k);
            d.p[i] = w;
            d.m[i] = m;
            d.v[i] = v;
            sw[nl * (SLAB_C * MAX_T) + r] = w;  // r = c_local * T + t
        }
    }
    if (d.nvar == 0) return;
    __syncthreads();
    const bool full_n = nn == SLAB_N && (N & 7) == 0;  // the transposed layouts take 8 output channels = 16 bytes per store
    const bool even_c = (cw & 1) == 0 && (C & 1) == 0;  // the forward layouts take 2 input channels = 4 bytes per store
    auto ldw = [&](int nl, int cl, int t) { return sw[nl * (SLAB_C * MAX_T) + cl * T + t]; };
    auto s2d_val = [&](int nl, int cl, int s, int t2, float sc) {
        const int sy = s >> 1, sx = s & 1, kh2 = t2 / 3, kw2 = t2 - kh2 * 3;
        if (kh2 < 2 && kw2 < 2 && (kh2 == 1 || sy == 1) && (kw2 == 1 || sx == 1)) {
            const int kh = kh2 == 0 ? 0 : 1 + sy, kw = kw2 == 0 ? 0 : 1 + sx;
            return sc * sw[nl * (SLAB_C * MAX_T) + cl * 9 + kh * 3 + kw];
        }
        return 0.f;
    };
    for (int vi = 0; vi < d.nvar; ++vi) {
        const int kind = d.var[vi].kind;
        const float sc = d.var[vi].scale;
        if (kind == STYLEX_ADAM_COPY_PACK) {  // [N][T][C] and / or [C][T][N], bf16
            unsigned short* wf = (unsigned short*)d.var[vi].a;
            unsigned short* wb = (unsigned short*)d.var[vi].b;
            if (wf && even_c) {
                const int ch = cw >> 1;
                for (int e = tid; e < nn * T * ch; e += 256) {
                    const int cl = (e % ch) << 1, t = (e / ch) % T, nl = e / (ch * T);
                    const unsigned v = (unsigned)bf16_rne(sc * ldw(nl, cl, t)) | ((unsigned)bf16_rne(sc * ldw(nl, cl + 1, t)) << 16);
                    *reinterpret_cast<unsigned*>(wf + ((int64_t)(n0 + nl) * T + t) * C + c0 + cl) = v;
                }
            } else if (wf) {
                for (int e = tid; e < nn * T * cw; e += 256) {
                    const int cl = e % cw, t = (e / cw) % T, nl = e / (cw * T);
                    wf[((int64_t)(n0 + nl) * T + t) * C + c0 + cl] = bf16_rne(sc * ldw(nl, cl, t));
                }
            }
            if (wb && full_n) {
                for (int e = tid; e < T * cw; e += 256) {
                    const int t = e % T, cl = e / T;
                    unsigned q[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        q[h] = (unsigned)bf16_rne(sc * ldw(2 * h, cl, t)) | ((unsigned)bf16_rne(sc * ldw(2 * h + 1, cl, t)) << 16);
                    *reinterpret_cast<uint4*>(wb + ((int64_t)(c0 + cl) * T + t) * N + n0) = make_uint4(q[0], q[1], q[2], q[3]);
                }
            } else if (wb) {
                for (int e = tid; e < nn * T * cw; e += 256) {
                    const int nl = e % nn, t = (e / nn) % T, cl = e / (nn * T);
                    wb[((int64_t)(c0 + cl) * T + t) * N + n0 + nl] = bf16_rne(sc * ldw(nl, cl, t));
                }
            }
        } else if (kind == STYLEX_ADAM_COPY_PACK_S2D) {  // space-to-depth form of a 3x3 / stride-2 conv (see pack_weight_s2d_kernel)
            unsigned short* wf = (unsigned short*)d.var[vi].a;
            unsigned short* wb = (unsigned short*)d.var[vi].b;
            const int C4 = 4 * C;
            if (wf && even_c) {
                const int ch = cw >> 1;
                for (int e = tid; e < nn * 36 * ch; e += 256) {
                    const int cl = (e % ch) << 1, s = (e / ch) & 3, t2 = (e / (ch * 4)) % 9, nl = e / (ch * 36);
                    const unsigned v = (unsigned)bf16_rne(s2d_val(nl, cl, s, t2, sc)) | ((unsigned)bf16_rne(s2d_val(nl, cl + 1, s, t2, sc)) << 16);
                    *reinterpret_cast<unsigned*>(wf + ((int64_t)(n0 + nl) * 9 + t2) * C4 + s * C + c0 + cl) = v;
                }
            } else if (wf) {
                for (int e = tid; e < nn * 36 * cw; e += 256) {
                    const int cl = e % cw, s = (e / cw) & 3, t2 = (e / (cw * 4)) % 9, nl = e / (cw * 36);
                    wf[((int64_t)(n0 + nl) * 9 + t2) * C4 + s * C + c0 + cl] = bf16_rne(s2d_val(nl, cl, s, t2, sc));
                }
            }
            if (wb && full_n) {
                for (int e = tid; e < 36 * cw; e += 256) {
                    const int t2 = e % 9, s = (e / 9) & 3, cl = e / 36;
                    unsigned q[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        q[h] = (unsigned)bf16_rne(s2d_val(2 * h, cl, s, t2, sc)) | ((unsigned)bf16_rne(s2d_val(2 * h + 1, cl, s, t2, sc)) << 16);
                    *reinterpret_cast<uint4*>(wb + ((int64_t)(s * C + c0 + cl) * 9 + t2) * N + n0) = make_uint4(q[0], q[1], q[2], q[3]);
                }
            } else if (wb) {
                for (int e = tid; e < nn * 36 * cw; e += 256) {
                    const int nl = e % nn, t2 = (e / nn) % 9, s = (e / (nn * 9)) & 3, cl = e / (nn * 36);
                    wb[((int64_t)(s * C + c0 + cl) * 9 + t2) * N + n0 + nl] = bf16_rne(s2d_val(nl, cl, s, t2, sc));
                }
            }
        } else if (kind == STYLEX_ADAM_COPY_SUMSQ) {  // wsq[n][c] = sum over the taps of w^2 (fp32)
            float* q = (float*)d.var[vi].a;
            for (int e = tid; e < nn * cw; e += 256) {
                const int cl = e % cw, nl = e / cw;
                float acc = 0.f;
                for (int t = 0; t < T; ++t) {
                    const float w = ldw(nl, cl, t);
                    acc += w * w;
                }
                q[(int64_t)(n0 + nl) * C + c0 + cl] = acc;
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t stylex_adam_pack_tensor_blocks(int64_t numel, int32_t N, int32_t C, int32_t T) {
    if (numel <= 0) return 0;
    if (N <= 0) return (numel + SLAB_ELEMS - 1) / SLAB_ELEMS;
    if (T < 1 || T > MAX_T || (int64_t)N * C * T != numel) return -1;
    return (int64_t)((N + SLAB_N - 1) / SLAB_N) * ((C + SLAB_C - 1) / SLAB_C);
}

int stylex_adam_pack_step(const stylex_adam_tensor* descs_dev, const int32_t* block_map_dev, int64_t n_blocks, void* stream) {
    if (!descs_dev || !block_map_dev || n_blocks < 0 || n_blocks > 0x7fffffff) return STYLEX_EINVAL;
    if (n_blocks == 0) return 0;
    hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, block_map_dev);
    return (int)hipGetLastError();
}

}  // extern "C"
