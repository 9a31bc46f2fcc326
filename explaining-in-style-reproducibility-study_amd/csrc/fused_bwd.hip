// fused_bwd.hip — backward-side elementwise + per-(image,channel) reduction kernels of the fused
// fast path (steps without gradient penalty / path-length regularisation need no double backward,
// so conv + bias/noise + LeakyReLU (+ demodulation / residual merge) run as ONE forward kernel and
// the gradient bookkeeping below replaces 5-8 separate full-tensor passes and reductions).
//
// Activation tensors NHWC fp32 or bf16 (act_dtype), sums in fp32.  One block = one image b x one contiguous pixel range; a thread owns VEC
// consecutive channels and walks pixels with stride rows-per-pass, accumulating up to 3 column sums
// in registers; the block combines them through LDS in fixed order and writes
// partial[b][chunk][k][C]; the (tiny) sum over chunks is done by the caller.  Deterministic.
//
// Reference ops covered (stylex/stylex_train.py): the backward of  x = lrelu(conv + bias) (:726-731),
// (x + res)/sqrt(2) (:743), x = lrelu(conv2dmod(x, style) + noise) (:700-714) and of the modulation
// / demodulation products inside Conv2DMod.forward (:650-656).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

namespace {

constexpr int NT = 256;

template <int K>
struct Acc {
    float4 v[K];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

// Block-level reduction over the pixel-row dimension and store of partial[b][chunk][k][c..c+3].
template <int K>
__device__ __forceinline__ void block_reduce_store(Acc<K>& a, float* smem, int cv, int prow, int CV, int rpp, int C,
                                                   float* partial_row /* [K][C] */, bool active) {
    // smem layout [K][rpp][CV] float4
    float4* s4 = reinterpret_cast<float4*>(smem);
    if (active) {
#pragma unroll
        for (int k = 0; k < K; ++k) s4[(k * rpp + prow) * CV + cv] = a.v[k];
    }
    __syncthreads();
    if (active && prow == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < rpp; ++r) {
                float4 u = s4[(k * rpp + r) * CV + cv];
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            float* d = partial_row + (long)k * C + cv * 4;
            if (cv * 4 + 3 < C) {
                *reinterpret_cast<float4*>(d) = t;
            } else {
                float tt[4] = {t.x, t.y, t.z, t.w};
                for (int e = 0; e < 4 && cv * 4 + e < C; ++e) d[e] = tt[e];
            }
        }
    }
}

struct Geo {
    int b, cv, prow, CV, rpp;
    long p_begin, p_end;  // pixel range of this block within image b
    bool active;
};

__device__ __forceinline__ Geo make_geo(int HW, int C, int nchunks) {
    Geo g;
    g.CV = (C + 3) / 4;
    if (g.CV > NT) g.CV = NT;  // C <= 1024 asserted by the host
    g.rpp = NT / g.CV;
    g.cv = threadIdx.x % g.CV;
    g.prow = threadIdx.x / g.CV;
    g.active = g.prow < g.rpp;
    g.b = blockIdx.y;
    long per = ((long)HW + nchunks - 1) / nchunks;
    g.p_begin = (long)blockIdx.x * per;
    g.p_end = g.p_begin + per < HW ? g.p_begin + per : HW;
    return g;
}

// activation loads/stores (fp32 or bf16 by runtime flag); C % 4 == 0 is checked on the host
__device__ __forceinline__ float4 ld4(const void* p, long off, int bf) { return act_ld4(p, off, bf); }
__device__ __forceinline__ void st4(void* p, long off, float4 v, int bf) { act_st4(p, off, v, bf); }
__device__ __forceinline__ float4 ldp4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// dx = dy * scale * (lrelu ? (y > 0 ? 1 : 0.2) : 1);  partial[b][chunk][c] = sum_pixels dx
__global__ __launch_bounds__(NT) void act_bwd_reduce_kernel(const void* __restrict__ dy, const void* __restrict__ y,
                                                           void* __restrict__ dx, float* __restrict__ partial, int HW,
                                                           int C, int nchunks, int lrelu, float scale, int bf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Geo g = make_geo(HW, C, nchunks);
    Acc<1> a;
    a.zero();
    const int c = g.cv * 4;
    if (g.active && c < C) {
        const long base = (long)g.b * HW;
        for (long p = g.p_begin + g.prow; p < g.p_end; p += g.rpp) {
            const long o = (base + p) * C + c;
            float4 gv = ld4(dy, o, bf);
            gv.x *= scale; gv.y *= scale; gv.z *= scale; gv.w *= scale;
            if (lrelu) {
                const float slope = lrelu == 2 ? 0.f : 0.2f;  // 2 = ReLU
                float4 yv = ld4(y, o, bf);
                gv.x = yv.x > 0.f ? gv.x : slope * gv.x;
                gv.y = yv.y > 0.f ? gv.y : slope * gv.y;
                gv.z = yv.z > 0.f ? gv.z : slope * gv.z;
                gv.w = yv.w > 0.f ? gv.w : slope * gv.w;
            }
            if (dx) st4(dx, o, gv, bf);
            a.v[0].x += gv.x; a.v[0].y += gv.y; a.v[0].z += gv.z; a.v[0].w += gv.w;
        }
    }
    block_reduce_store<1>(a, smem, g.cv, g.prow, g.CV, g.rpp, C, partial + ((long)g.b * nchunks + blockIdx.x) * C,
                          g.active && c < C);
}

// Backward prologue of  y = lrelu( d[b,c] * z + noise[b,w,h]*nw[c] + nb[c] ):
//   gz = gy * lrelu'(y)                       (written; the conv gradients apply d while staging)
//   S0[b,c] = sum gz * (d*z)   with d*z = lrelu^-1(y) - noise   => grad wrt d is S0 / d
//   S1[b,c] = sum gz * noise_plane[b,w,h]     => grad wrt noise weight (summed over b by the caller)
//   S2[b,c] = sum gz                          => grad wrt noise bias
__global__ __launch_bounds__(NT) void modconv_bwd_prep_kernel(const void* __restrict__ gy, const void* __restrict__ y,
                                                             const float* __restrict__ noise, int ns,
                                                             const float* __restrict__ nw, const float* __restrict__ nb,
                                                             void* __restrict__ gz, float* __restrict__ partial, int H,
                                                             int W, int C, int nchunks, int lrelu, int bf,
                                                             const float* __restrict__ gz_scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = H * W;
    Geo g = make_geo(HW, C, nchunks);
    Acc<3> a;
    a.zero();
    const int c = g.cv * 4;
    if (g.active && c < C) {
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = w4;
        if (noise) {
            w4 = ldp4(nw + c);
            b4 = ldp4(nb + c);
        }
        float4 d4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (gz_scale) d4 = ldp4(gz_scale + (long)g.b * C + c);
        const long base = (long)g.b * HW;
        for (long p = g.p_begin + g.prow; p < g.p_end; p += g.rpp) {
            const long o = (base + p) * C + c;
            float4 gv = ld4(gy, o, bf);
            float4 yv = ld4(y, o, bf);
            float nz = 0.f;
            if (noise) {
                const int pi = (int)p, h = pi / W, w = pi - h * W;  // p < H*W: a 32-bit division (the 64-bit one cost ~100 VALU ops per pixel)
                nz = noise[((long)g.b * ns + w) * ns + h];
            }
            float4 t = yv;  // pre-activation
            if (lrelu) {
                gv.x = yv.x > 0.f ? gv.x : 0.2f * gv.x; t.x = yv.x > 0.f ? yv.x : 5.f * yv.x;
                gv.y = yv.y > 0.f ? gv.y : 0.2f * gv.y; t.y = yv.y > 0.f ? yv.y : 5.f * yv.y;
                gv.z = yv.z > 0.f ? gv.z : 0.2f * gv.z; t.z = yv.z > 0.f ? yv.z : 5.f * yv.z;
                gv.w = yv.w > 0.f ? gv.w : 0.2f * gv.w; t.w = yv.w > 0.f ? yv.w : 5.f * yv.w;
            }
            // the STORED gradient optionally carries the demodulation coefficient d[b][c] (both consumers — the data
            // gradient and the weight gradient — want gz * d; folding it here makes their operands scale-free, so the
            // data gradient can take the LDS-DMA kernel); the sums below use the unscaled value
            if (gz_scale) st4(gz, o, make_float4(gv.x * d4.x, gv.y * d4.y, gv.z * d4.z, gv.w * d4.w), bf);
            else st4(gz, o, gv, bf);
            a.v[0].x += gv.x * (t.x - (nz * w4.x + b4.x));
            a.v[0].y += gv.y * (t.y - (nz * w4.y + b4.y));
            a.v[0].z += gv.z * (t.z - (nz * w4.z + b4.z));
            a.v[0].w += gv.w * (t.w - (nz * w4.w + b4.w));
            a.v[1].x += gv.x * nz; a.v[1].y += gv.y * nz; a.v[1].z += gv.z * nz; a.v[1].w += gv.w * nz;
            a.v[2].x += gv.x; a.v[2].y += gv.y; a.v[2].z += gv.z; a.v[2].w += gv.w;
        }
    }
    block_reduce_store<3>(a, smem, g.cv, g.prow, g.CV, g.rpp, C, partial + ((long)g.b * nchunks + blockIdx.x) * 3 * C,
                          g.active && c < C);
}

// gx = t * s[b,c];  partial[b][chunk][c] = sum_pixels x * t     (grad wrt the modulation scale)
__global__ __launch_bounds__(NT) void scale_reduce_kernel(const void* __restrict__ x, const void* __restrict__ t,
                                                         const float* __restrict__ s, void* __restrict__ gx,
                                                         float* __restrict__ partial, int HW, int C, int nchunks, int bf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    Geo g = make_geo(HW, C, nchunks);
    Acc<1> a;
    a.zero();
    const int c = g.cv * 4;
    if (g.active && c < C) {
        float4 s4 = ldp4(s + (long)g.b * C + c);
        const long base = (long)g.b * HW;
        for (long p = g.p_begin + g.prow; p < g.p_end; p += g.rpp) {
            const long o = (base + p) * C + c;
            float4 tv = ld4(t, o, bf);
            float4 xv = ld4(x, o, bf);
            a.v[0].x += xv.x * tv.x; a.v[0].y += xv.y * tv.y; a.v[0].z += xv.z * tv.z; a.v[0].w += xv.w * tv.w;
            if (gx) {
                tv.x *= s4.x; tv.y *= s4.y; tv.z *= s4.z; tv.w *= s4.w;
                st4(gx, o, tv, bf);
            }
        }
    }
    block_reduce_store<1>(a, smem, g.cv, g.prow, g.CV, g.rpp, C, partial + ((long)g.b * nchunks + blockIdx.x) * C,
                          g.active && c < C);
}

inline size_t reduce_smem(int C, int K) {
    int CV = (C + 3) / 4;
    if (CV > NT) CV = NT;
    int rpp = NT / CV;
    return (size_t)K * rpp * CV * 16;
}

inline bool ok_shape(const int64_t* sh, int nchunks) {
    return sh[0] > 0 && sh[1] > 0 && sh[2] > 0 && sh[3] > 0 && sh[3] <= 1024 && (sh[3] % 4 == 0) && nchunks > 0 &&
           sh[0] <= 65535;
}

}  // namespace

extern "C" {

int stylex_reduce_chunks(const int64_t* sh) {
    // blocks ~ 2048 in total, at least 64 pixels per block
    long HW = (long)sh[1] * sh[2];
    long n = (2048 + sh[0] - 1) / sh[0];
    long maxc = (HW + 63) / 64;
    if (n > maxc) n = maxc;
    if (n < 1) n = 1;
    return (int)n;
}

int stylex_act_bwd_reduce(const void* dy, const void* y, void* dx, float* partial, const int64_t* sh, int nchunks,
                          int lrelu, float scale, int act_dtype, void* stream) {
    if (act_dtype != 0 && act_dtype != 1) return STYLEX_EINVAL;
    if (!dy || !partial || !ok_shape(sh, nchunks) || (lrelu && !y)) return STYLEX_EINVAL;
    int HW = (int)(sh[1] * sh[2]), C = (int)sh[3];
    hipLaunchKernelGGL(act_bwd_reduce_kernel, dim3(nchunks, (unsigned)sh[0]), dim3(NT), reduce_smem(C, 1),
                       (hipStream_t)stream, dy, y, dx, partial, HW, C, nchunks, lrelu, scale, act_dtype);
    return (int)hipGetLastError();
}

static int modconv_bwd_prep_impl(const void* gy, const void* y, const float* noise, int64_t noise_stride,
                                 const float* noise_w, const float* noise_b, void* gz, float* partial, const int64_t* sh,
                                 int nchunks, int lrelu, int act_dtype, const float* gz_scale, void* stream);

int stylex_modconv_bwd_prep(const void* gy, const void* y, const float* noise, int64_t noise_stride,
                            const float* noise_w, const float* noise_b, void* gz, float* partial, const int64_t* sh,
                            int nchunks, int lrelu, int act_dtype, void* stream) {
    return modconv_bwd_prep_impl(gy, y, noise, noise_stride, noise_w, noise_b, gz, partial, sh, nchunks, lrelu, act_dtype,
                                 nullptr, stream);
}

int stylex_modconv_bwd_prep_scaled(const void* gy, const void* y, const float* noise, int64_t noise_stride,
                                   const float* noise_w, const float* noise_b, const float* gz_scale, void* gz,
                                   float* partial, const int64_t* sh, int nchunks, int lrelu, int act_dtype, void* stream) {
    if (!gz_scale || (reinterpret_cast<uintptr_t>(gz_scale) & 15)) return STYLEX_EINVAL;
    return modconv_bwd_prep_impl(gy, y, noise, noise_stride, noise_w, noise_b, gz, partial, sh, nchunks, lrelu, act_dtype,
                                 gz_scale, stream);
}

static int modconv_bwd_prep_impl(const void* gy, const void* y, const float* noise, int64_t noise_stride,
                                 const float* noise_w, const float* noise_b, void* gz, float* partial, const int64_t* sh,
                                 int nchunks, int lrelu, int act_dtype, const float* gz_scale, void* stream) {
    if (act_dtype != 0 && act_dtype != 1) return STYLEX_EINVAL;
    if (!gy || !y || !gz || !partial || !ok_shape(sh, nchunks)) return STYLEX_EINVAL;
    if (noise && (!noise_w || !noise_b || noise_stride < sh[1] || noise_stride < sh[2])) return STYLEX_EINVAL;
    int C = (int)sh[3];
    hipLaunchKernelGGL(modconv_bwd_prep_kernel, dim3(nchunks, (unsigned)sh[0]), dim3(NT), reduce_smem(C, 3),
                       (hipStream_t)stream, gy, y, noise, (int)noise_stride, noise_w, noise_b, gz, partial, (int)sh[1],
                       (int)sh[2], C, nchunks, lrelu, act_dtype, gz_scale);
    return (int)hipGetLastError();
}

int stylex_scale_reduce(const void* x, const void* t, const float* s, void* gx, float* partial, const int64_t* sh,
                        int nchunks, int act_dtype, void* stream) {
    if (act_dtype != 0 && act_dtype != 1) return STYLEX_EINVAL;
    if (!x || !t || !s || !partial || !ok_shape(sh, nchunks)) return STYLEX_EINVAL;
    int HW = (int)(sh[1] * sh[2]), C = (int)sh[3];
    hipLaunchKernelGGL(scale_reduce_kernel, dim3(nchunks, (unsigned)sh[0]), dim3(NT), reduce_smem(C, 1),
                       (hipStream_t)stream, x, t, s, gx, partial, HW, C, nchunks, act_dtype);
    return (int)hipGetLastError();
}

}  // extern "C"
