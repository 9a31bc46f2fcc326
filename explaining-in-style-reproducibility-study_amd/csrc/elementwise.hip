// elementwise.hip — HBM-bound kernels of the StylEx step on gfx950: bilinear x2 and its adjoint,
// 3x3 reflect blur and its adjoint, bias(+noise)+LeakyReLU and its gradient mask, row-wise
// squared norms, even-pixel gather.  Tensors NHWC, fp32 or bf16 (act_dtype); one lane handles VEC consecutive
// channels of a pixel (4 fp32 / 8 bf16 = 16 bytes, or 1 for odd channel counts) so that a wave reads/writes whole
// lines.  Grid-stride loops.  bf16 blurs take the column-strip kernel below (5 TB/s vs 2-3 for the direct form).
//
// Reference ops replaced (file stylex/stylex_train.py): nn.Upsample(scale_factor=2, bilinear,
// align_corners=False) :614,679; Blur :144-153 (kornia filter2d, reflect); nn.Conv2d bias +
// leaky_relu(0.2) :340-341,726-731; noise add :696-714; gradients.norm(2, dim=1) :302.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

namespace {

template <int V>
struct Vec;
template <>
struct Vec<4> {
    typedef float4 T;
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ T ld(const void* p, long off, int bf) { return act_ld4(p, off, bf); }
    static __device__ __forceinline__ void st(void* p, long off, T v, int bf) { act_st4(p, off, v, bf); }
    static __device__ __forceinline__ void fma(T& a, float w, T x) {
        a.x = fmaf(w, x.x, a.x); a.y = fmaf(w, x.y, a.y); a.z = fmaf(w, x.z, a.z); a.w = fmaf(w, x.w, a.w);
    }
};
template <>
struct Vec<1> {
    typedef float T;
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ T ld(const void* p, long off, int bf) { return act_ld1(p, off, bf); }
    static __device__ __forceinline__ void st(void* p, long off, T v, int bf) { act_st1(p, off, v, bf); }
    static __device__ __forceinline__ void fma(T& a, float w, T x) { a = fmaf(w, x, a); }
};

struct F8 {
    float4 a, b;
};
template <>
struct Vec<8> {  // bf16 activations only: 8 channels = one 16-byte access per lane
    typedef F8 T;
    static __device__ __forceinline__ T zero() { return F8{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}; }
    static __device__ __forceinline__ T ld(const void* p, long off, int) {
        uint4 h = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + off);
        return F8{act_unpack4(make_uint2(h.x, h.y)), act_unpack4(make_uint2(h.z, h.w))};
    }
    static __device__ __forceinline__ void st(void* p, long off, T v, int) {
        uint2 lo = act_pack4(v.a), hi = act_pack4(v.b);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p) + off) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    static __device__ __forceinline__ void fma(T& acc, float w, T x) {
        Vec<4>::fma(acc.a, w, x.a);
        Vec<4>::fma(acc.b, w, x.b);
    }
};

// ---- bilinear x2 index rule (exact integer arithmetic) -------------------------------------
// out[2k]   = .25*in[max(k-1,0)] + .75*in[k]
// out[2k+1] = .75*in[k]          + .25*in[min(k+1,n-1)]
__device__ __forceinline__ void up_rule(int o, int n, int& lo, int& hi, float& w_hi) {
    int k = o >> 1;
    if (o & 1) {
        lo = k;
        hi = min(k + 1, n - 1);
        w_hi = 0.25f;
    } else {
        lo = max(k - 1, 0);
        hi = k;
        w_hi = 0.75f;
    }
}
// weight with which output o reads input i
__device__ __forceinline__ float up_coef(int o, int i, int n) {
    int lo, hi;
    float wh;
    up_rule(o, n, lo, hi, wh);
    return (lo == i ? 1.f - wh : 0.f) + (hi == i ? wh : 0.f);
}

// i = ((b * Hd + h) * Wd + w) * cv + cq  ->  (cq, w, h, b); returns the pixel index i / cv.  32-bit divisions whenever
// the flat index fits (always, for the StylEx shapes): a 64-bit division by a run-time value is ~100 VALU instructions
// and the five of them per 16-byte vector made the resampling kernels issue-bound, not HBM-bound.
__device__ __forceinline__ long decomp_index(long i, int cv, int Wd, int Hd, int& cq, int& w, int& h, int& b) {
    if (i < (1L << 31)) {
        const unsigned u = (unsigned)i, pix = u / (unsigned)cv, r = pix / (unsigned)Wd;
        cq = (int)(u - pix * (unsigned)cv);
        w = (int)(pix - r * (unsigned)Wd);
        b = (int)(r / (unsigned)Hd);
        h = (int)(r - (unsigned)b * (unsigned)Hd);
        return (long)pix;
    }
    const long pix = i / cv, r = pix / Wd;
    cq = (int)(i - pix * cv);
    w = (int)(pix - r * Wd);
    b = (int)(r / Hd);
    h = (int)(r - (long)b * Hd);
    return pix;
}

template <int V>
__global__ void upsample2x_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int H, int W, int C, int bf) {
    const int cv = C / V;
    const long total = (long)B * 2 * H * 2 * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, ow, oh, b;
        const long pix = decomp_index(i, cv, (2 * W), (2 * H), cq, ow, oh, b);
        const int c = cq * V;
        int hl, hh, wl, wh_;
        float fh, fw;
        up_rule(oh, H, hl, hh, fh);
        up_rule(ow, W, wl, wh_, fw);
        const long base = (long)b * H * W * C + c;
        typename Vec<V>::T acc = Vec<V>::zero();
        // same association as the reference kernel: h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11)
        typename Vec<V>::T r0 = Vec<V>::zero(), r1 = Vec<V>::zero();
        Vec<V>::fma(r0, 1.f - fw, Vec<V>::ld(x, base + ((long)hl * W + wl) * C, bf));
        Vec<V>::fma(r0, fw, Vec<V>::ld(x, base + ((long)hl * W + wh_) * C, bf));
        Vec<V>::fma(r1, 1.f - fw, Vec<V>::ld(x, base + ((long)hh * W + wl) * C, bf));
        Vec<V>::fma(r1, fw, Vec<V>::ld(x, base + ((long)hh * W + wh_) * C, bf));
        Vec<V>::fma(acc, 1.f - fh, r0);
        Vec<V>::fma(acc, fh, r1);
        Vec<V>::st(y, pix * C + c, acc, bf);
    }
}

template <int V>
__global__ void upsample2x_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx, int B, int H, int W, int C, int bf) {
    const int cv = C / V;
    const long total = (long)B * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, iw, ih, b;
        const long pix = decomp_index(i, cv, W, H, cq, iw, ih, b);
        const int c = cq * V;
        const long base = (long)b * 4 * H * W * C + c;
        typename Vec<V>::T acc = Vec<V>::zero();
#pragma unroll
        for (int a = -1; a <= 2; ++a) {
            int oh = 2 * ih + a;
            if (oh < 0 || oh >= 2 * H) continue;
            float ch = up_coef(oh, ih, H);
            if (ch == 0.f) continue;
#pragma unroll
            for (int e = -1; e <= 2; ++e) {
                int ow = 2 * iw + e;
                if (ow < 0 || ow >= 2 * W) continue;
                float cw = up_coef(ow, iw, W);
                if (cw == 0.f) continue;
                Vec<V>::fma(acc, ch * cw, Vec<V>::ld(dy, base + ((long)oh * 2 * W + ow) * C, bf));
            }
        }
        Vec<V>::st(dx, pix * C + c, acc, bf);
    }
}

// ---- 3x3 binomial blur, reflect border: index rule -1 -> 1, n -> n-2 -------------------------
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// ---- RGB skip path of a GeneratorBlock in one pass: out = blur3x3_reflect(upsample2x(rgb + prev)) ------------------
// (RGBBlock.forward, reference stylex_train.py:618-629: skip add :622-623, nn.Upsample :614 + Blur :615 as
// self.upsample :625-626).  Both resamplers are separable and linear, so the chain is one 3x3 stencil over the
// LOW-resolution sum whose row / column weights depend on the output parity and on the borders:
//   ub_coef(o, i, n) = sum_{d=-1..1} (1,2,1)[d]/4 * up_coef(reflect1(o + d, 2n), i, n),  non-zero for i in {k-1, k, k+1},
// k = o >> 1.  The adjoint gathers the 6x6 outputs whose stencils cover an input pixel.
__device__ __forceinline__ float ub_coef(int o, int i, int n) {
    const int n2 = 2 * n;
    return 0.25f * up_coef(reflect1(o - 1, n2), i, n) + 0.5f * up_coef(o, i, n) + 0.25f * up_coef(reflect1(o + 1, n2), i, n);
}

template <int V>
__global__ void rgb_up_blur_add_fwd_kernel(const void* __restrict__ rgb, const void* __restrict__ prev, void* __restrict__ y,
                                           int B, int H, int W, int C, int bf) {
    const int cv = C / V;
    const long total = (long)B * 2 * H * 2 * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, ow, oh, b;
        const long pix = decomp_index(i, cv, 2 * W, 2 * H, cq, ow, oh, b);
        const int c = cq * V, kh = oh >> 1, kw = ow >> 1;
        const long base = (long)b * H * W * C + c;
        float cw[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int iw = kw - 1 + e;
            cw[e] = (iw >= 0 && iw < W) ? ub_coef(ow, iw, W) : 0.f;
        }
        typename Vec<V>::T acc = Vec<V>::zero();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int ih = kh - 1 + a;
            if (ih < 0 || ih >= H) continue;
            const float ch = ub_coef(oh, ih, H);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (cw[e] == 0.f) continue;
                const long off = base + ((long)ih * W + (kw - 1 + e)) * C;
                typename Vec<V>::T t = Vec<V>::ld(rgb, off, bf);
                if (prev) Vec<V>::fma(t, 1.f, Vec<V>::ld(prev, off, bf));
                Vec<V>::fma(acc, ch * cw[e], t);
            }
        }
        Vec<V>::st(y, pix * C + c, acc, bf);
    }
}

template <int V>
__global__ void rgb_up_blur_add_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx, int B, int H, int W, int C,
                                           int bf) {
    const int cv = C / V;
    const long total = (long)B * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, iw, ih, b;
        const long pix = decomp_index(i, cv, W, H, cq, iw, ih, b);
        const int c = cq * V;
        const long base = (long)b * 4 * H * W * C + c;
        float cw[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int ow = 2 * iw - 2 + e;
            cw[e] = (ow >= 0 && ow < 2 * W) ? ub_coef(ow, iw, W) : 0.f;
        }
        typename Vec<V>::T acc = Vec<V>::zero();
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const int oh = 2 * ih - 2 + a;
            if (oh < 0 || oh >= 2 * H) continue;
            const float ch = ub_coef(oh, ih, H);
            if (ch == 0.f) continue;
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                if (cw[e] == 0.f) continue;
                Vec<V>::fma(acc, ch * cw[e], Vec<V>::ld(dy, base + ((long)oh * 2 * W + (2 * iw - 2 + e)) * C, bf));
            }
        }
        Vec<V>::st(dx, pix * C + c, acc, bf);
    }
}


// space-to-depth address of element (b, h, w, c) of a [B,H,W,C] tensor stored as [B,H/2,W/2,4C]
__device__ __forceinline__ long s2d_off(int b, int h, int w, int c, int H, int W, int C) {
    return (((long)b * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * (4L * C) + (((h & 1) * 2 + (w & 1)) * C) + c;
}

template <int V>
__global__ void blur3x3_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int H, int W, int C, int bf,
                                   int s2d) {
    const int cv = C / V;
    const long total = (long)B * H * W * cv;
    const float f[3] = {1.f, 2.f, 1.f};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, w, h, b;
        const long pix = decomp_index(i, cv, W, H, cq, w, h, b);
        const int c = cq * V;
        const long base = (long)b * H * W * C + c;
        typename Vec<V>::T acc = Vec<V>::zero();
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            int hh = reflect1(h + dh, H);
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                int ww = reflect1(w + dw, W);
                Vec<V>::fma(acc, f[dh + 1] * f[dw + 1] * (1.f / 16.f), Vec<V>::ld(x, base + ((long)hh * W + ww) * C, bf));
            }
        }
        Vec<V>::st(y, s2d ? s2d_off(b, h, w, c, H, W, C) : pix * C + c, acc, bf);
    }
}

// adjoint weight: sum_d f[d] * [reflect(o+d) == i]
__device__ __forceinline__ float blur_coef(int o, int i, int n) {
    float s = 0.f;
    if (reflect1(o - 1, n) == i) s += 1.f;
    if (o == i) s += 2.f;
    if (reflect1(o + 1, n) == i) s += 1.f;
    return s;
}

template <int V>
__global__ void blur3x3_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx, int B, int H, int W, int C, int bf,
                                   int s2d, const void* __restrict__ gate, float gslope) {
    const int cv = C / V;
    const long total = (long)B * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, w, h, b;
        const long pix = decomp_index(i, cv, W, H, cq, w, h, b);
        const int c = cq * V;
        const long base = (long)b * H * W * C + c;
        typename Vec<V>::T acc = Vec<V>::zero();
#pragma unroll
        for (int a = -1; a <= 1; ++a) {
            int oh = h + a;
            if (oh < 0 || oh >= H) continue;
            float ch = blur_coef(oh, h, H);
#pragma unroll
            for (int e = -1; e <= 1; ++e) {
                int ow = w + e;
                if (ow < 0 || ow >= W) continue;
                float cw = blur_coef(ow, w, W);
                Vec<V>::fma(acc, ch * cw * (1.f / 16.f),
                            Vec<V>::ld(dy, s2d ? s2d_off(b, oh, ow, c, H, W, C) : base + ((long)oh * W + ow) * C, bf));
            }
        }
        if (gate) {  // derivative of the (Leaky)ReLU that produced the blurred tensor, fused into the adjoint
            float a[V], g[V];
            *reinterpret_cast<typename Vec<V>::T*>(a) = acc;
            *reinterpret_cast<typename Vec<V>::T*>(g) = Vec<V>::ld(gate, pix * C + c, bf);
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] = g[e] > 0.f ? a[e] : gslope * a[e];
            acc = *reinterpret_cast<typename Vec<V>::T*>(a);
        }
        Vec<V>::st(dx, pix * C + c, acc, bf);
    }
}

// ---- stride-2 pixel subsampling and its adjoint (zero insertion) --------------------------------
// A 1x1 / stride-2 convolution (DiscriminatorBlock.conv_res) is a 1x1 / stride-1 convolution of the even
// pixels: gathering them once turns its three GEMMs into contiguous ones, and the data gradient no longer runs
// a full-resolution implicit GEMM whose rows are 3/4 structural zeros.  shape = the FULL-resolution [B,H,W,C].
template <int V>
__global__ void subsample2_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int H, int W, int C, int bf) {
    const int cv = C / V, Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
    const long total = (long)B * Ho * Wo * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, ow, oh, b;
        const long pix = decomp_index(i, cv, Wo, Ho, cq, ow, oh, b);
        const int c = cq * V;
        Vec<V>::st(y, pix * C + c, Vec<V>::ld(x, (((long)b * H + 2 * oh) * W + 2 * ow) * C + c, bf), bf);
    }
}

template <int V>
__global__ void subsample2_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx, int B, int H, int W, int C, int bf) {
    const int cv = C / V, Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
    const long total = (long)B * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, w, h, b;
        const long pix = decomp_index(i, cv, W, H, cq, w, h, b);
        const int c = cq * V;
        typename Vec<V>::T v = Vec<V>::zero();
        if (!((h | w) & 1)) v = Vec<V>::ld(dy, (((long)b * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c, bf);
        Vec<V>::st(dx, pix * C + c, v, bf);
    }
}

// dst[b, 2i, 2j, :] += src[b, i, j, :]  (in place; dst full resolution [B,H,W,C], src [B,ceil(H/2),ceil(W/2),C]).
// The merge of a DiscriminatorBlock's two input gradients: the 3x3 path's data gradient (dst) and the adjoint of the
// even-pixel gather of the 1x1/stride-2 residual conv (zero insertion of src) — without materialising the
// zero-inserted tensor and without a full-resolution add.
template <int V>
__global__ void add_at_even_kernel(const void* __restrict__ src, void* __restrict__ dst, int B, int H, int W, int C, int bf) {
    const int cv = C / V, Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
    const long total = (long)B * Ho * Wo * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, ow, oh, b;
        const long pix = decomp_index(i, cv, Wo, Ho, cq, ow, oh, b);
        const int c = cq * V;
        const long o = (((long)b * H + 2 * oh) * W + 2 * ow) * C + c;
        float a[V], d[V];
        *reinterpret_cast<typename Vec<V>::T*>(a) = Vec<V>::ld(src, pix * C + c, bf);
        *reinterpret_cast<typename Vec<V>::T*>(d) = Vec<V>::ld(dst, o, bf);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] += a[e];
        Vec<V>::st(dst, o, *reinterpret_cast<typename Vec<V>::T*>(d), bf);
    }
}

// ---- bf16 column-strip blur (forward and adjoint) ---------------------------------------------
// One lane owns 8 channels of one column and walks ROWS output rows with a 3-row sliding window of
// horizontally filtered values, so every input row is fetched once per strip (+2 halo rows) instead of three
// times; the x+-1 neighbours are the lines the adjacent lanes fetch anyway.  Used for bf16 tensors only (the
// fp32 parity mode keeps the direct 9-tap kernel and its summation order).
template <bool ADJ>
__device__ __forceinline__ void blur_tap(int pos, int d, int n, int& idx, float& coef) {
    if (!ADJ) {
        idx = reflect1(pos + d, n);
        coef = d == 0 ? 2.f : 1.f;
    } else {
        idx = pos + d;
        if (idx < 0 || idx >= n) {
            idx = pos;
            coef = 0.f;
        } else {
            coef = blur_coef(idx, pos, n);
        }
    }
}

template <bool ADJ, int ROWS>
__global__ __launch_bounds__(256) void blur3x3_strip_kernel(const unsigned short* __restrict__ in,
                                                            unsigned short* __restrict__ out, int B, int H, int W, int C,
                                                            int s2d, const unsigned short* __restrict__ gate, float gslope,
                                                            const unsigned char* __restrict__ gmask) {
    const int cv = C >> 3;
    const int strips = (H + ROWS - 1) / ROWS;
    const long total = (long)B * strips * W * cv;
    const bool in_s2d = ADJ && s2d, out_s2d = !ADJ && s2d;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, w, hs, b;
        decomp_index(i, cv, W, strips, cq, w, hs, b);
        const int c = cq * 8, h0 = hs * ROWS;
        int iw[3];
        float cw[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) blur_tap<ADJ>(w, e - 1, W, iw[e], cw[e]);
        auto addr = [&](int hh, int ww) -> long {
            return in_s2d ? s2d_off(b, hh, ww, c, H, W, C) : (((long)b * H + hh) * W + ww) * C + c;
        };
        auto hrow = [&](int hh) -> F8 {  // horizontally filtered row hh (already a valid row index)
            F8 acc = Vec<8>::zero();
#pragma unroll
            for (int e = 0; e < 3; ++e) Vec<8>::fma(acc, cw[e], Vec<8>::ld(in, addr(hh, iw[e]), 1));
            return acc;
        };
        int ih;
        float ch;
        F8 win[3];
        // gate bits of the strip's rows, fetched up front (addresses are known; a load issued in the row loop right
        // before its use would expose its latency ROWS times)
        unsigned gm[ROWS];
        if (ADJ && gmask) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                gm[r] = (h0 + r < H) ? gmask[((((long)b * H + h0 + r) * W + w) * C + c) >> 3] : 0u;
        }
        blur_tap<ADJ>(h0, -1, H, ih, ch);
        win[0] = hrow(ih);
        win[1] = hrow(h0);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int h = h0 + r;
            if (h >= H) break;
            float cv3[3];
            int dummy;
            blur_tap<ADJ>(h, 1, H, ih, cv3[2]);
            win[2] = hrow(ih);
            blur_tap<ADJ>(h, -1, H, dummy, cv3[0]);
            cv3[1] = ADJ ? blur_coef(h, h, H) : 2.f;
            F8 acc = Vec<8>::zero();
#pragma unroll
            for (int a = 0; a < 3; ++a) Vec<8>::fma(acc, cv3[a] * (1.f / 16.f), win[a]);
            const long o = out_s2d ? s2d_off(b, h, w, c, H, W, C) : (((long)b * H + h) * W + w) * C + c;
            if (ADJ && gate) {  // activation derivative fused into the adjoint (see blur3x3_bwd_kernel)
                float a[8], g[8];
                *reinterpret_cast<F8*>(a) = acc;
                *reinterpret_cast<F8*>(g) = Vec<8>::ld(gate, o, 1);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = g[e] > 0.f ? a[e] : gslope * a[e];
                acc = *reinterpret_cast<F8*>(a);
            } else if (ADJ && gmask) {  // the gate as one bit per element (STYLEX_EPI_MASK_OUT of the forward conv)
                float a[8];
                *reinterpret_cast<F8*>(a) = acc;
                const unsigned m = gm[r];
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = ((m >> e) & 1u) ? a[e] : gslope * a[e];
                acc = *reinterpret_cast<F8*>(a);
            }
            Vec<8>::st(out, o, acc, 1);
            win[0] = win[1];
            win[1] = win[2];
        }
    }
}

// ---- bias (+ transposed noise) + LeakyReLU(0.2) ----------------------------------------------
template <int V>
__global__ void bias_act_fwd_kernel(const void* __restrict__ x, const float* __restrict__ bias,
                                    const float* __restrict__ noise, long ns, const float* __restrict__ nw,
                                    const float* __restrict__ nb, void* __restrict__ y, int B, int H, int W, int C,
                                    int bf) {
    const int cv = C / V;
    const long total = (long)B * H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cq, w, h, b;
        const long pix = decomp_index(i, cv, W, H, cq, w, h, b);
        const int c = cq * V;
        float nz = 0.f;
        if (noise) nz = noise[((long)b * ns + w) * ns + h];  // (sic) spatially transposed, stylex_train.py:696-698
        float v[V];
        *reinterpret_cast<typename Vec<V>::T*>(v) = Vec<V>::ld(x, pix * C + c, bf);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = v[e];
            if (bias) t += bias[c + e];
            if (noise) t += fmaf(nz, nw[c + e], nb[c + e]);
            v[e] = t > 0.f ? t : 0.2f * t;
        }
        Vec<V>::st(y, pix * C + c, *reinterpret_cast<typename Vec<V>::T*>(v), bf);
    }
}

template <int V>
__global__ void bias_act_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ y, void* __restrict__ dx,
                                    long n, int bf) {
    const long total = n / V;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float g[V], o[V];
        *reinterpret_cast<typename Vec<V>::T*>(g) = Vec<V>::ld(dy, i * V, bf);
        *reinterpret_cast<typename Vec<V>::T*>(o) = Vec<V>::ld(y, i * V, bf);
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = o[e] > 0.f ? g[e] : 0.2f * g[e];
        Vec<V>::st(dx, i * V, *reinterpret_cast<typename Vec<V>::T*>(g), bf);
    }
}

// ---- row-wise sum of squares: wave shuffle -> LDS -> one value per row (fixed order) ----------
__global__ __launch_bounds__(1024) void rowwise_sumsq_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             long cols) {
    __shared__ float part[16];
    const float* row = x + (long)blockIdx.x * cols;
    float s = 0.f;
    const bool vec = (cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
    if (vec) {
        for (long i = threadIdx.x; i < cols / 4; i += blockDim.x) {
            float4 v = *reinterpret_cast<const float4*>(row + i * 4);
            s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
        }
    } else {
        for (long i = threadIdx.x; i < cols; i += blockDim.x) s = fmaf(row[i], row[i], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
        out[blockIdx.x] = t;
    }
}

inline int grid_for(long work) {
    long b = (work + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}
inline bool vec_ok(int C, const void* a, const void* b) {
    return (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0) && ((reinterpret_cast<uintptr_t>(b) & 15) == 0);
}

}  // namespace

#define LAUNCH_EW(kern, work, a, b, ...)                                                                    \
    do {                                                                                                    \
        if (bf && (C % 8 == 0) && vec_ok(C, a, b))                                                          \
            hipLaunchKernelGGL(kern<8>, dim3(grid_for((work) / 8)), dim3(256), 0, s, __VA_ARGS__);          \
        else if (vec_ok(C, a, b))                                                                           \
            hipLaunchKernelGGL(kern<4>, dim3(grid_for((work) / 4)), dim3(256), 0, s, __VA_ARGS__);          \
        else                                                                                                \
            hipLaunchKernelGGL(kern<1>, dim3(grid_for(work)), dim3(256), 0, s, __VA_ARGS__);                \
        return (int)hipGetLastError();                                                                      \
    } while (0)

static bool blur_strip_ok(int bf, int H, int C, const void* a, const void* b) {
    return bf && (C % 8 == 0) && H >= 8 && ((reinterpret_cast<uintptr_t>(a) & 15) == 0) &&
           ((reinterpret_cast<uintptr_t>(b) & 15) == 0);
}
template <bool ADJ>
static int launch_blur_strip(const void* in, void* out, int B, int H, int W, int C, int s2d, hipStream_t s,
                             const void* gate = nullptr, float gslope = 0.f, const void* gmask = nullptr) {
    constexpr int ROWS = 8;
    long work = (long)B * ((H + ROWS - 1) / ROWS) * W * (C / 8);
    long blocks = (work + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((blur3x3_strip_kernel<ADJ, ROWS>), dim3((unsigned)blocks), dim3(256), 0, s,
                       (const unsigned short*)in, (unsigned short*)out, B, H, W, C, s2d, (const unsigned short*)gate,
                       gslope, (const unsigned char*)gmask);
    return (int)hipGetLastError();
}

extern "C" {

#define EW_ARGS const int64_t* sh, int act_dtype, void* stream
#define EW_UNPACK                                                     \
    hipStream_t s = (hipStream_t)stream;                               \
    int B = (int)sh[0], H = (int)sh[1], W = (int)sh[2], C = (int)sh[3]; \
    int bf = act_dtype == 1;                                           \
    if (act_dtype != 0 && act_dtype != 1) return STYLEX_EINVAL;

int stylex_upsample2x_bilinear_fwd(const void* x, void* y, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return STYLEX_EINVAL;
    LAUNCH_EW(upsample2x_fwd_kernel, (long)B * 4 * H * W * C, x, y, x, y, B, H, W, C, bf);
}
int stylex_upsample2x_bilinear_bwd(const void* dy, void* dx, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return STYLEX_EINVAL;
    LAUNCH_EW(upsample2x_bwd_kernel, (long)B * H * W * C, dy, dx, dy, dx, B, H, W, C, bf);
}
int stylex_rgb_up_blur_add_fwd(const void* rgb, const void* prev, void* y, EW_ARGS) {
    EW_UNPACK
    if (!rgb || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return STYLEX_EINVAL;
    const void* al = prev ? prev : rgb;
    // (no 8-channel instantiation: the RGB path has 4 channels per pixel, and the SLP vectoriser gave that variant packed fp32
    // fmas with a crossed operand — the form DESIGN §7a keeps out of every translation unit)
    if (vec_ok(C, rgb, y) && vec_ok(C, al, y))
        hipLaunchKernelGGL(rgb_up_blur_add_fwd_kernel<4>, dim3(grid_for((long)B * 4 * H * W * C / 4)), dim3(256), 0, s, rgb, prev,
                           y, B, H, W, C, bf);
    else
        hipLaunchKernelGGL(rgb_up_blur_add_fwd_kernel<1>, dim3(grid_for((long)B * 4 * H * W * C)), dim3(256), 0, s, rgb, prev, y,
                           B, H, W, C, bf);
    return (int)hipGetLastError();
}
int stylex_rgb_up_blur_add_bwd(const void* dy, void* dx, EW_ARGS) {
    EW_UNPACK
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return STYLEX_EINVAL;
    LAUNCH_EW(rgb_up_blur_add_bwd_kernel, (long)B * H * W * C, dy, dx, dy, dx, B, H, W, C, bf);
}
int stylex_blur3x3_reflect_fwd(const void* x, void* y, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H < 2 || W < 2 || C <= 0) return STYLEX_EINVAL;
    if (blur_strip_ok(bf, H, C, x, y)) return launch_blur_strip<false>(x, y, B, H, W, C, 0, s);
    LAUNCH_EW(blur3x3_fwd_kernel, (long)B * H * W * C, x, y, x, y, B, H, W, C, bf, 0);
}
static int blur_bwd_impl(const void* dy, const void* gate, float gslope, void* dx, int s2d, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H < 2 || W < 2 || C <= 0 || (s2d && ((H & 1) || (W & 1)))) return STYLEX_EINVAL;
    const bool gate_ok = !gate || (reinterpret_cast<uintptr_t>(gate) & 15) == 0;
    if (gate_ok && blur_strip_ok(bf, H, C, dy, dx)) return launch_blur_strip<true>(dy, dx, B, H, W, C, s2d, s, gate, gslope);
    if (gate && !gate_ok) {
        hipLaunchKernelGGL(blur3x3_bwd_kernel<1>, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, s, dy, dx, B, H, W, C, bf,
                           s2d, gate, gslope);
        return (int)hipGetLastError();
    }
    LAUNCH_EW(blur3x3_bwd_kernel, (long)B * H * W * C, dy, dx, dy, dx, B, H, W, C, bf, s2d, gate, gslope);
}
int stylex_blur3x3_reflect_bwd(const void* dy, void* dx, EW_ARGS) { return blur_bwd_impl(dy, nullptr, 0.f, dx, 0, sh, act_dtype, stream); }
int stylex_blur3x3_reflect_bwd_gate(const void* dy, const void* gate, float slope, void* dx, EW_ARGS) {
    if (!gate) return STYLEX_EINVAL;
    return blur_bwd_impl(dy, gate, slope, dx, 0, sh, act_dtype, stream);
}
int stylex_blur3x3_s2d_bwd_gate(const void* dy, const void* gate, float slope, void* dx, EW_ARGS) {
    if (!gate) return STYLEX_EINVAL;
    return blur_bwd_impl(dy, gate, slope, dx, 1, sh, act_dtype, stream);
}
int stylex_blur3x3_s2d_bwd_gate_mask(const void* dy, const void* mask, float slope, void* dx, EW_ARGS) {
    EW_UNPACK
    if (!dy || !mask || !dx || B <= 0 || H < 2 || W < 2 || C <= 0 || (H & 1) || (W & 1)) return STYLEX_EINVAL;
    if (!blur_strip_ok(bf, H, C, dy, dx)) return STYLEX_EINVAL;  // the strip kernel is the only reader of masks
    return launch_blur_strip<true>(dy, dx, B, H, W, C, 1, s, nullptr, slope, mask);
}
int stylex_blur3x3_s2d_fwd(const void* x, void* y, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H < 2 || W < 2 || C <= 0 || (H & 1) || (W & 1)) return STYLEX_EINVAL;
    if (blur_strip_ok(bf, H, C, x, y)) return launch_blur_strip<false>(x, y, B, H, W, C, 1, s);
    LAUNCH_EW(blur3x3_fwd_kernel, (long)B * H * W * C, x, y, x, y, B, H, W, C, bf, 1);
}
int stylex_blur3x3_s2d_bwd(const void* dy, void* dx, EW_ARGS) { return blur_bwd_impl(dy, nullptr, 0.f, dx, 1, sh, act_dtype, stream); }
// RGB image (3 channels, any strides, fp32 or bf16) -> one 16-byte channel slot per pixel: bf16 NHWC [B][H][W][8], channels
// 3..7 zero.  One launch instead of cast + channels_last copy + zero tensor + cat in front of the first conv of the
// discriminator / encoder (DiscriminatorBlock 0 reads RGB; the vector load paths want whole 16-byte slots).
__global__ __launch_bounds__(256) void pad_rgb8_kernel(const void* __restrict__ x, uint4* __restrict__ y, long npix, int H, int W,
                                                       long sb, long sc, long sh, long sw, int x_bf16) {
    for (long i = blockIdx.x * 256l + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W);
        const long t = i / W;
        const int h = (int)(t % H);
        const long b = t / H;
        const long o = b * sb + h * sh + w * sw;
        unsigned short c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (x_bf16) c[k] = reinterpret_cast<const unsigned short*>(x)[o + k * sc];
            else c[k] = (unsigned short)(act_pack2(reinterpret_cast<const float*>(x)[o + k * sc], 0.f) & 0xffffu);
        }
        y[i] = make_uint4((unsigned)c[0] | ((unsigned)c[1] << 16), (unsigned)c[2], 0u, 0u);
    }
}

int stylex_pad_rgb8(const void* x, void* y, const int64_t* shape, const int64_t* strides, int x_is_bf16, void* stream) {
    if (!x || !y || !shape || !strides || shape[0] < 1 || shape[1] < 1 || shape[2] < 1) return STYLEX_EINVAL;
    if (reinterpret_cast<uintptr_t>(y) & 15) return STYLEX_EINVAL;
    const long npix = (long)shape[0] * shape[1] * shape[2];
    long blocks = (npix + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pad_rgb8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (uint4*)y, npix, (int)shape[1],
                       (int)shape[2], (long)strides[0], (long)strides[1], (long)strides[2], (long)strides[3], x_is_bf16);
    return (int)hipGetLastError();
}
int stylex_subsample2_fwd(const void* x, void* y, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return STYLEX_EINVAL;
    LAUNCH_EW(subsample2_fwd_kernel, (long)B * ((H + 1) / 2) * ((W + 1) / 2) * C, x, y, x, y, B, H, W, C, bf);
}
int stylex_subsample2_bwd(const void* dy, void* dx, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return STYLEX_EINVAL;
    LAUNCH_EW(subsample2_bwd_kernel, (long)B * H * W * C, dy, dx, dy, dx, B, H, W, C, bf);
}
int stylex_add_at_even(const void* src, void* dst, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || !src || !dst) return STYLEX_EINVAL;
    LAUNCH_EW(add_at_even_kernel, (long)B * ((H + 1) / 2) * ((W + 1) / 2) * C, src, dst, src, dst, B, H, W, C, bf);
}
int stylex_bias_act_fwd(const void* x, const float* bias, const float* noise, int64_t noise_stride,
                        const float* noise_w, const float* noise_b, void* y, EW_ARGS) {
    EW_UNPACK
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return STYLEX_EINVAL;
    if (noise && (!noise_w || !noise_b || noise_stride < H || noise_stride < W)) return STYLEX_EINVAL;
    LAUNCH_EW(bias_act_fwd_kernel, (long)B * H * W * C, x, y, x, bias, noise, (long)noise_stride, noise_w, noise_b, y, B,
              H, W, C, bf);
}
int stylex_bias_act_bwd(const void* dy, const void* y, void* dx, EW_ARGS) {
    EW_UNPACK
    (void)B; (void)H; (void)W; (void)C;
    long n = (long)sh[0] * sh[1] * sh[2] * sh[3];
    if (n <= 0) return STYLEX_EINVAL;
    bool v = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
             ((reinterpret_cast<uintptr_t>(dx) & 15) == 0);
    if (v)
        hipLaunchKernelGGL(bias_act_bwd_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, s, dy, y, dx, n, bf);
    else
        hipLaunchKernelGGL(bias_act_bwd_kernel<1>, dim3(grid_for(n)), dim3(256), 0, s, dy, y, dx, n, bf);
    return (int)hipGetLastError();
}
int stylex_rowwise_sumsq(const float* x, float* out, const int64_t* sh, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (sh[0] <= 0 || sh[1] <= 0) return STYLEX_EINVAL;
    hipLaunchKernelGGL(rowwise_sumsq_kernel, dim3((unsigned)sh[0]), dim3(1024), 0, s, x, out, (long)sh[1]);
    return (int)hipGetLastError();
}

}  // extern "C"
