// torgb.hip — the generator's to-RGB layer as streaming kernels (bf16 NHWC activations).
//
// Reference: RGBBlock.forward (stylex/stylex_train.py:618-621) = Conv2DMod(input_channel, 3, kernel 1, demod=False)
// (:611, :647-667):   y[b,p,n] = sum_c x[b,p,c] * (style[b,c] + 1) * W[n,c],   n < 3.
//
// As a GEMM this layer has 3 output columns: on the MFMA path it spends a 32-wide tile on them and three launches
// (data gradient, x*t reduction, weight gradient) on its backward — 0.36 / 0.46 / 0.46 ms at 256 px, batch 64, for
// 268 MB of input (0.75 TB/s).  It is a pure HBM stream, so it is written as one:
//   forward : read x once, write 4 bf16 per pixel (the fourth is zero: 8-byte pixels keep the consumers' vector paths)
//   backward: read x and gy once, write gx, and reduce T[b,n,c] = sum_p x[b,p,c] * gy[b,p,n] in registers; the caller
//             derives  d style = sum_n W[n,c] T[b,n,c]  and  dW[n,c] = sum_b (style+1)[b,c] T[b,n,c]  from the
//             (B x 3 x C) result.  One launch replaces the three above.
// A lane owns 8 consecutive channels (one 16-byte load) of a pixel; the LP = C/8 lanes of a pixel combine their three
// partial dot products with wave shuffles.  grid = (pixel chunks, B): per-image products W*(style+1) live in registers.
// Deterministic: fixed shuffle / LDS reduction order, partial[b][chunk][3][C] summed by the caller.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

namespace {

constexpr int NT = 256;

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    f32x2_t t = {a, b};
    bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
    return *reinterpret_cast<unsigned*>(&r);
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(u[i] << 16);
        f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
}

__device__ __forceinline__ void load_products(const float* s1, const float* w, int b, int C, int cg, float (&m)[3][8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s = s1[(long)b * C + cg * 8 + e];
#pragma unroll
        for (int n = 0; n < 3; ++n) m[n][e] = w[n * C + cg * 8 + e] * s;
    }
}

__global__ __launch_bounds__(NT) void torgb_fwd_kernel(const uint4* __restrict__ x, const float* __restrict__ s1,
                                                      const float* __restrict__ w, uint2* __restrict__ y, int HW, int C,
                                                      int lp_shift, int px_per_block) {
    const int b = blockIdx.y, LP = 1 << lp_shift, tid = threadIdx.x;
    const int cg = tid & (LP - 1), pl = tid >> lp_shift, PPB = NT >> lp_shift;
    float m[3][8];
    load_products(s1, w, b, C, cg, m);
    const int p0 = blockIdx.x * px_per_block;
    const int p1 = min(p0 + px_per_block, HW);
    const uint4* xb = x + (long)b * HW * LP;
    uint2* yb = y + (long)b * HW;
    for (int p = p0 + pl; p < p1; p += PPB) {  // the LP lanes of a pixel share p: they stay converged for the shuffles
        float f[8];
        unpack8(xb[(long)p * LP + cg], f);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a0 += f[e] * m[0][e];
            a1 += f[e] * m[1][e];
            a2 += f[e] * m[2][e];
        }
        for (int o = LP >> 1; o; o >>= 1) {
            a0 += __shfl_xor(a0, o);
            a1 += __shfl_xor(a1, o);
            a2 += __shfl_xor(a2, o);
        }
        if (cg == 0) yb[p] = make_uint2(pack2(a0, a1), pack2(a2, 0.f));
    }
}

template <bool WANT_GX>
__global__ __launch_bounds__(NT) void torgb_bwd_kernel(const uint4* __restrict__ x, const uint2* __restrict__ gy,
                                                      const float* __restrict__ s1, const float* __restrict__ w,
                                                      uint4* __restrict__ gx, float* __restrict__ partial, int HW, int C,
                                                      int lp_shift, int px_per_block) {
    __shared__ float red[NT / 64][64][25];  // [wave][lane][3*8 sums], padded against bank conflicts
    const int b = blockIdx.y, LP = 1 << lp_shift, tid = threadIdx.x;
    const int cg = tid & (LP - 1), pl = tid >> lp_shift, PPB = NT >> lp_shift;
    float m[3][8];
    if (WANT_GX) load_products(s1, w, b, C, cg, m);
    float acc[3][8];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[n][e] = 0.f;
    const int p0 = blockIdx.x * px_per_block;
    const int p1 = min(p0 + px_per_block, HW);
    const uint4* xb = x + (long)b * HW * LP;
    uint4* gxb = gx + (long)b * HW * LP;
    const uint2* gyb = gy + (long)b * HW;
#pragma unroll 2
    for (int p = p0 + pl; p < p1; p += PPB) {
        const uint2 g = gyb[p];
        const float g0 = __uint_as_float(g.x << 16), g1 = __uint_as_float(g.x & 0xffff0000u),
                    g2 = __uint_as_float(g.y << 16);
        float f[8];
        unpack8(xb[(long)p * LP + cg], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[0][e] += f[e] * g0;
            acc[1][e] += f[e] * g1;
            acc[2][e] += f[e] * g2;
        }
        if (WANT_GX) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = g0 * m[0][e] + g1 * m[1][e] + g2 * m[2][e];
            gxb[(long)p * LP + cg] = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
        }
    }
    // lanes of one wave that own the same channel group (lane % LP) -> lane < LP, then the four waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
    for (int o = LP; o < 64; o <<= 1) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[n][e] += __shfl_xor(acc[n][e], o);
    }
    if (lane < LP) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][lane][n * 8 + e] = acc[n][e];
    }
    __syncthreads();
    if (tid < LP) {
        float* dst = partial + ((long)b * gridDim.x + blockIdx.x) * 3 * C;
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = red[0][tid][n * 8 + e];
                for (int wv = 1; wv < NT / 64; ++wv) t += red[wv][tid][n * 8 + e];
                dst[n * C + tid * 8 + e] = t;
            }
    }
}

inline int lp_shift_of(int C) {
    if (C < 8 || C > 512 || (C & (C - 1))) return -1;
    int s = 0;
    while ((8 << s) < C) ++s;
    return s;
}

}  // namespace

extern "C" {

int stylex_torgb_chunks(const int64_t* sh) {
    // a block pass covers 2048 / C pixels; give every block >= 8 passes (the per-image products and the block
    // reduction are amortised over them) and stop splitting at ~4096 blocks in total
    const long HW = (long)sh[1] * sh[2], C = sh[3] < 8 ? 8 : sh[3];
    const long per_pass = 2048 / C > 0 ? 2048 / C : 1;
    long n = (HW + 8 * per_pass - 1) / (8 * per_pass);
    const long cap = 4096 / (sh[0] > 0 ? sh[0] : 1);
    if (n > cap) n = cap;
    return (int)(n < 1 ? 1 : n);
}

int stylex_torgb_fwd(const void* x, const float* s1, const float* w, void* y, const int64_t* sh, void* stream) {
    if (!x || !s1 || !w || !y || sh[0] <= 0 || sh[0] > 65535 || sh[1] <= 0 || sh[2] <= 0) return STYLEX_EINVAL;
    const int C = (int)sh[3], ls = lp_shift_of(C);
    if (ls < 0) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 7)) return STYLEX_EINVAL;
    const int HW = (int)(sh[1] * sh[2]), nch = stylex_torgb_chunks(sh);
    const int ppb = (HW + nch - 1) / nch;
    hipLaunchKernelGGL(torgb_fwd_kernel, dim3(nch, (unsigned)sh[0]), dim3(NT), 0, (hipStream_t)stream,
                       (const uint4*)x, s1, w, (uint2*)y, HW, C, ls, ppb);
    return (int)hipGetLastError();
}

int stylex_torgb_bwd(const void* x, const void* gy, const float* s1, const float* w, void* gx, float* partial,
                     const int64_t* sh, void* stream) {
    if (!x || !gy || !s1 || !w || !partial || sh[0] <= 0 || sh[0] > 65535 || sh[1] <= 0 || sh[2] <= 0)
        return STYLEX_EINVAL;
    const int C = (int)sh[3], ls = lp_shift_of(C);
    if (ls < 0) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(gx) & 15) ||
        (reinterpret_cast<uintptr_t>(gy) & 7))
        return STYLEX_EINVAL;
    const int HW = (int)(sh[1] * sh[2]), nch = stylex_torgb_chunks(sh);
    const int ppb = (HW + nch - 1) / nch;
    if (gx)
        hipLaunchKernelGGL(torgb_bwd_kernel<true>, dim3(nch, (unsigned)sh[0]), dim3(NT), 0, (hipStream_t)stream,
                           (const uint4*)x, (const uint2*)gy, s1, w, (uint4*)gx, partial, HW, C, ls, ppb);
    else
        hipLaunchKernelGGL(torgb_bwd_kernel<false>, dim3(nch, (unsigned)sh[0]), dim3(NT), 0, (hipStream_t)stream,
                           (const uint4*)x, (const uint2*)gy, s1, w, (uint4*)gx, partial, HW, C, ls, ppb);
    return (int)hipGetLastError();
}

}  // extern "C"
