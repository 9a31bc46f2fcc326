// frozen_ew.hip — the elementwise tail of the frozen classifier's conv layers (reference
// stylex/resnet_classifier.py:29-71: a torchvision ResNet-18 in eval mode, weights without gradients).
//
// The convolutions of the classifier stay on the stock library (fp32, north_star); what runs between them does not have
// to be four library launches per conv.  An eval-mode BatchNorm is the per-channel affine map y = x * s[c] + t[c]
// (s = gamma / sqrt(var + eps), t = beta - mean * s), so `bn -> relu`, `bn -> (+ identity) -> relu` and the stem's
// `bn -> relu -> maxpool(3, 2, 1)` are each ONE pass over the conv output here, with a first-order backward towards the
// input image (the generated batch is classified with gradients, reference stylex_train.py:1452-1459).  HBM-bound fp32
// NCHW streams: per step they were ~2.5 ms of library kernels (BatchNorm 60 launches, ReLU 51, add 24, max-pool 3+1,
// threshold_backward 17, BatchNorm backward 20+20).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

namespace {

// y = act(x * s[c] + t[c] (+ r)); V consecutive elements of one (b, c) plane per lane
template <int V>
__global__ __launch_bounds__(256) void affine_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                             const float* __restrict__ t, const float* __restrict__ r,
                                                             float* __restrict__ y, unsigned total_v, unsigned HW, unsigned C,
                                                             int relu) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total_v; i += gridDim.x * 256u) {
        const unsigned e = i * V;
        const unsigned c = (e / HW) % C;
        const float sc = s[c], sh = t[c];
        float v[V], rr[V];
        if (V == 4) {
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(x + e);
            if (r) *reinterpret_cast<float4*>(rr) = *reinterpret_cast<const float4*>(r + e);
        } else {
            v[0] = x[e];
            if (r) rr[0] = r[e];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float o = fmaf(v[k], sc, sh);
            if (r) o += rr[k];
            v[k] = relu ? fmaxf(o, 0.f) : o;
        }
        if (V == 4) *reinterpret_cast<float4*>(y + e) = *reinterpret_cast<float4*>(v);
        else y[e] = v[0];
    }
}

// gres = gy * [y > 0] (when relu), gx = gres * s[c]; gres may be null (no residual), gx may alias nothing else
template <int V>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                             const float* __restrict__ s, float* __restrict__ gx,
                                                             float* __restrict__ gres, unsigned total_v, unsigned HW, unsigned C,
                                                             int relu) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total_v; i += gridDim.x * 256u) {
        const unsigned e = i * V;
        const unsigned c = (e / HW) % C;
        const float sc = s[c];
        float g[V], yy[V], o[V];
        if (V == 4) {
            *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(gy + e);
            if (relu) *reinterpret_cast<float4*>(yy) = *reinterpret_cast<const float4*>(y + e);
        } else {
            g[0] = gy[e];
            if (relu) yy[0] = y[e];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (relu && !(yy[k] > 0.f)) g[k] = 0.f;
            o[k] = g[k] * sc;
        }
        if (V == 4) {
            *reinterpret_cast<float4*>(gx + e) = *reinterpret_cast<float4*>(o);
            if (gres) *reinterpret_cast<float4*>(gres + e) = *reinterpret_cast<float4*>(g);
        } else {
            gx[e] = o[0];
            if (gres) gres[e] = g[0];
        }
    }
}

// y[b][c][oh][ow] = max over the 3x3 / stride-2 / pad-1 window of relu(x * s[c] + t[c]); idx = window-local position of
// the maximum (row-major scan, first strict maximum — ATen's rule), 255 when the maximum is not positive (the ReLU then
// blocks the gradient whichever element the pool would have picked).
__global__ __launch_bounds__(256) void affine_relu_maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                                      const float* __restrict__ t, float* __restrict__ y,
                                                                      unsigned char* __restrict__ idx, unsigned total, int C,
                                                                      int H, int W, int Ho, int Wo) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int ow = i % Wo, oh = (i / Wo) % Ho;
        const unsigned plane = i / (Wo * Ho);
        const int c = plane % C;
        const float sc = s[c], sh = t[c];
        const float* xp = x + (size_t)plane * H * W;
        float best = -INFINITY;
        int bi = 255;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = 2 * oh - 1 + kh;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = 2 * ow - 1 + kw;
                if (w < 0 || w >= W) continue;
                const float v = fmaf(xp[h * W + w], sc, sh);
                if (v > best) {
                    best = v;
                    bi = kh * 3 + kw;
                }
            }
        }
        y[i] = fmaxf(best, 0.f);
        if (idx) idx[i] = best > 0.f ? (unsigned char)bi : (unsigned char)255;
    }
}

// gx[b][c][h][w] = s[c] * sum of gy over the (<= 4) windows whose recorded maximum is this element
__global__ __launch_bounds__(256) void affine_relu_maxpool_bwd_kernel(const float* __restrict__ gy,
                                                                      const unsigned char* __restrict__ idx,
                                                                      const float* __restrict__ s, float* __restrict__ gx,
                                                                      unsigned total, int C, int H, int W, int Ho, int Wo) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int w = i % W, h = (i / W) % H;
        const unsigned plane = i / (W * H);
        const int c = plane % C;
        const float* gp = gy + (size_t)plane * Ho * Wo;
        const unsigned char* ip = idx + (size_t)plane * Ho * Wo;
        float acc = 0.f;
        // windows oh with 2*oh - 1 <= h <= 2*oh + 1
        const int oh0 = h >> 1, oh1 = (h + 1) >> 1, ow0 = w >> 1, ow1 = (w + 1) >> 1;
        for (int oh = oh0; oh <= oh1; ++oh) {
            if (oh >= Ho) continue;
            const int kh = h - (2 * oh - 1);
            for (int ow = ow0; ow <= ow1; ++ow) {
                if (ow >= Wo) continue;
                const int kw = w - (2 * ow - 1);
                if (ip[oh * Wo + ow] == kh * 3 + kw) acc += gp[oh * Wo + ow];
            }
        }
        gx[i] = acc * s[c];
    }
}

// ---- LPIPS distance of one feature tap (lpips 0.1.4, lpips.py: normalize_tensor + spatial_average(lin(diff^2))) -------
//   n0 = f0 / (sqrt(sum_c f0^2) + 1e-10), n1 likewise;  d[b][p] = sum_c lin[c] * (n0 - n1)^2;  out[b] = mean_p d[b][p]
// One lane = one pixel of one sample; channels are strided by HW (coalesced across lanes).  Two channel sweeps (norms,
// then the weighted squared difference — the reference's arithmetic, no expanded-square cancellation); the second
// sweep hits L2 / MALL.  Per-block sums of d / HW go to partial[b][block] (summed once over all taps by the caller, in
// fixed order); the norms are kept for the backward.
constexpr float LPIPS_EPS = 1e-10f;
constexpr int LP_PIX = 64;  // pixels per block = one wave width; the block's 4 waves split the channels

// Sum over the block's SL waves (channel slices) of a per-lane value, in fixed order; every lane of every wave gets it.
template <int SL>
__device__ __forceinline__ float lp_cross_wave(float v, float (*buf)[LP_PIX], int wave, int lane) {
    buf[wave][lane] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < SL; k += 4) r += (buf[k][lane] + buf[k + 1][lane]) + (buf[k + 2][lane] + buf[k + 3][lane]);
    __syncthreads();
    return r;
}

// grid (ceil(HW / 64), B), block 64 * SL: lane = pixel (coalesced along HW), wave w takes channels w, w + SL, ...; the channel
// loops are unrolled 8x so that 16 independent loads are in flight per lane (a one-lane-per-pixel loop over 384 channels
// was pure load latency: 360 us per tap at 15x15).  SL = 4, or 16 for the small taps (round 6: the 15 x 15 taps are 128 blocks
// whatever the block size — with 4 waves each lane walked 96 channels in 12 dependent round trips, ~130 us per launch; 16
// waves make it 3).
template <int SL>
__global__ __launch_bounds__(64 * SL) void lpips_tap_fwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                            const float* __restrict__ lin, float* __restrict__ partial,
                                                            float* __restrict__ r0, float* __restrict__ r1, int C, int HW,
                                                            long partial_stride) {
    __shared__ float buf[SL][LP_PIX];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * LP_PIX + lane;
    const bool ok = p < HW;
    const float* a = f0 + (size_t)b * C * HW + (ok ? p : 0);
    const float* q = f1 + (size_t)b * C * HW + (ok ? p : 0);
    float s0 = 0.f, s1 = 0.f;
    int c = wave;
    for (; c + 7 * SL < C; c += 8 * SL) {
        float x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + SL * u) * HW];
            y[u] = q[(size_t)(c + SL * u) * HW];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s0 = fmaf(x[u], x[u], s0);
            s1 = fmaf(y[u], y[u], s1);
        }
    }
    for (; c < C; c += SL) {
        const float x = a[(size_t)c * HW], y = q[(size_t)c * HW];
        s0 = fmaf(x, x, s0);
        s1 = fmaf(y, y, s1);
    }
    s0 = lp_cross_wave<SL>(s0, buf, wave, lane);
    s1 = lp_cross_wave<SL>(s1, buf, wave, lane);
    const float n0 = sqrtf(s0), n1 = sqrtf(s1);
    if (ok && wave == 0) {
        if (r0) r0[(size_t)b * HW + p] = n0;
        if (r1) r1[(size_t)b * HW + p] = n1;
    }
    const float i0 = 1.f / (n0 + LPIPS_EPS), i1 = 1.f / (n1 + LPIPS_EPS);
    float d = 0.f;
    c = wave;
    for (; c + 7 * SL < C; c += 8 * SL) {
        float x[8], y[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + SL * u) * HW];
            y[u] = q[(size_t)(c + SL * u) * HW];
            l[u] = lin[c + SL * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float t = x[u] * i0 - y[u] * i1;
            d = fmaf(l[u] * t, t, d);
        }
    }
    for (; c < C; c += SL) {
        const float t = a[(size_t)c * HW] * i0 - q[(size_t)c * HW] * i1;
        d = fmaf(lin[c] * t, t, d);
    }
    d = lp_cross_wave<SL>(ok ? d : 0.f, buf, wave, lane);
    if (wave == 0) {  // fixed-order sum over the block's 64 pixels
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (lane == 0) partial[(size_t)b * partial_stride + blockIdx.x] = d / (float)HW;
    }
}

// d out[b] / d f1[c][p] = g[b] / HW * ( q_c / B - (sum_k q_k y_k) / (B^2 r1) * y_c ),  q_c = -2 lin_c (x_c / A - y_c / B),
// A = r0 + eps, B = r1 + eps  (and symmetrically for f0).  A pixel whose features are all zero gives 0 / 0 = NaN, as the
// reference's sqrt backward does.  Same thread layout as the forward.
template <int SL>
__global__ __launch_bounds__(64 * SL) void lpips_tap_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                            const float* __restrict__ lin, const float* __restrict__ r0,
                                                            const float* __restrict__ r1, const float* __restrict__ gout,
                                                            float* __restrict__ g0, float* __restrict__ g1, int C, int HW) {
    __shared__ float buf[SL][LP_PIX];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * LP_PIX + lane;
    const bool ok = p < HW;
    const size_t base = (size_t)b * C * HW + (ok ? p : 0);
    const float* a = f0 + base;
    const float* q = f1 + base;
    const float n0 = r0[(size_t)b * HW + (ok ? p : 0)], n1 = r1[(size_t)b * HW + (ok ? p : 0)];
    const float A = n0 + LPIPS_EPS, Bn = n1 + LPIPS_EPS;
    const float iA = 1.f / A, iB = 1.f / Bn;
    float dot0 = 0.f, dot1 = 0.f;  // sum_c lin_c t_c x_c, sum_c lin_c t_c y_c  (t = n0 - n1)
    int c = wave;
    for (; c + 7 * SL < C; c += 8 * SL) {
        float x[8], y[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + SL * u) * HW];
            y[u] = q[(size_t)(c + SL * u) * HW];
            l[u] = lin[c + SL * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float lt = l[u] * (x[u] * iA - y[u] * iB);
            dot0 = fmaf(lt, x[u], dot0);
            dot1 = fmaf(lt, y[u], dot1);
        }
    }
    for (; c < C; c += SL) {
        const float x = a[(size_t)c * HW], y = q[(size_t)c * HW];
        const float lt = lin[c] * (x * iA - y * iB);
        dot0 = fmaf(lt, x, dot0);
        dot1 = fmaf(lt, y, dot1);
    }
    dot0 = lp_cross_wave<SL>(dot0, buf, wave, lane);
    dot1 = lp_cross_wave<SL>(dot1, buf, wave, lane);
    if (!ok) return;
    const float gs = gout[b] / (float)HW;
    // d/dx_c = 2 lin_c t_c / A - 2 dot0 / (A^2 n0) x_c ;  d/dy_c = -2 lin_c t_c / B + 2 dot1 / (B^2 n1) y_c
    const float k0 = 2.f * dot0 / (A * A * n0), k1 = 2.f * dot1 / (Bn * Bn * n1);
    c = wave;
    for (; c + 7 * SL < C; c += 8 * SL) {
        float x[8], y[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + SL * u) * HW];
            y[u] = q[(size_t)(c + SL * u) * HW];
            l[u] = lin[c + SL * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float lt2 = 2.f * l[u] * (x[u] * iA - y[u] * iB);
            if (g0) g0[base + (size_t)(c + SL * u) * HW] = gs * (lt2 * iA - k0 * x[u]);
            if (g1) g1[base + (size_t)(c + SL * u) * HW] = gs * (k1 * y[u] - lt2 * iB);
        }
    }
    for (; c < C; c += SL) {
        const float x = a[(size_t)c * HW], y = q[(size_t)c * HW];
        const float lt2 = 2.f * lin[c] * (x * iA - y * iB);
        if (g0) g0[base + (size_t)c * HW] = gs * (lt2 * iA - k0 * x);
        if (g1) g1[base + (size_t)c * HW] = gs * (k1 * y - lt2 * iB);
    }
}

// ---- the same tap on bf16 NHWC features (round 6: LPIPS-AlexNet on this library's bf16 conv kernels in the speed mode) ----------
// f[b][p][c]: a pixel's channels are contiguous, so the channel reductions that cost the NCHW kernels a strided sweep are 16-byte
// loads here.  8 lanes per pixel (lane s of the octet takes the 8-channel slots s, s + 8, ...: 128 contiguous bytes per octet and
// load), 8 pixels per wave, 32 per block; octet sums by DPP-free xor shuffles in fixed order.  Same arithmetic as the NCHW
// kernels (two sweeps, no expanded square); partial[b][block] = sum of the block's 32 pixels / HW, block = 0 .. ceil(HW / 32) - 1.
constexpr int LPN_PIX = 32;

__device__ __forceinline__ void lp_unpack8(uint4 v, float* o) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        o[2 * d] = __uint_as_float(w[d] << 16);
        o[2 * d + 1] = __uint_as_float(w[d] & 0xffff0000u);
    }
}
__device__ __forceinline__ unsigned lp_pack2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 t = {a, b};
    bf2 r = __builtin_convertvector(t, bf2);
    return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ float lp_octet_sum(float v) {  // over the 8 lanes of a pixel, fixed order
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__global__ __launch_bounds__(256) void lpips_tap_nhwc_fwd_kernel(const unsigned short* __restrict__ f0,
                                                                 const unsigned short* __restrict__ f1,
                                                                 const float* __restrict__ lin, float* __restrict__ partial,
                                                                 float* __restrict__ r0, float* __restrict__ r1, int C, int HW,
                                                                 long partial_stride) {
    __shared__ float wsum[4];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 7, p = blockIdx.x * LPN_PIX + wave * 8 + (lane >> 3);
    const bool ok = p < HW;
    const size_t base = ((size_t)b * HW + (ok ? p : 0)) * C;
    const int slots = C >> 3;
    float s0 = 0.f, s1 = 0.f;
    for (int q = s; q < slots; q += 8) {
        float x[8], y[8];
        lp_unpack8(*reinterpret_cast<const uint4*>(f0 + base + q * 8), x);
        lp_unpack8(*reinterpret_cast<const uint4*>(f1 + base + q * 8), y);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s0 = fmaf(x[e], x[e], s0);
            s1 = fmaf(y[e], y[e], s1);
        }
    }
    s0 = lp_octet_sum(s0);
    s1 = lp_octet_sum(s1);
    const float n0 = sqrtf(s0), n1 = sqrtf(s1);
    if (ok && s == 0) {
        if (r0) r0[(size_t)b * HW + p] = n0;
        if (r1) r1[(size_t)b * HW + p] = n1;
    }
    const float i0 = 1.f / (n0 + LPIPS_EPS), i1 = 1.f / (n1 + LPIPS_EPS);
    float d = 0.f;
    for (int q = s; q < slots; q += 8) {
        float x[8], y[8];
        lp_unpack8(*reinterpret_cast<const uint4*>(f0 + base + q * 8), x);
        lp_unpack8(*reinterpret_cast<const uint4*>(f1 + base + q * 8), y);
        const float4 l0 = *reinterpret_cast<const float4*>(lin + q * 8), l1 = *reinterpret_cast<const float4*>(lin + q * 8 + 4);
        const float l[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = x[e] * i0 - y[e] * i1;
            d = fmaf(l[e] * t, t, d);
        }
    }
    d = lp_octet_sum(ok ? d : 0.f);
    // the wave's 8 pixels, then the block's 4 waves, in fixed order
    d += __shfl_xor(d, 8, 64);
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    if (lane == 0) wsum[wave] = d;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * partial_stride + blockIdx.x] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / (float)HW;
}

// gradients as lpips_tap_bwd_kernel (same formulas), written bf16 NHWC
__global__ __launch_bounds__(256) void lpips_tap_nhwc_bwd_kernel(const unsigned short* __restrict__ f0,
                                                                 const unsigned short* __restrict__ f1,
                                                                 const float* __restrict__ lin, const float* __restrict__ r0,
                                                                 const float* __restrict__ r1, const float* __restrict__ gout,
                                                                 unsigned short* __restrict__ g0, unsigned short* __restrict__ g1,
                                                                 int C, int HW) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 7, p = blockIdx.x * LPN_PIX + wave * 8 + (lane >> 3);
    const bool ok = p < HW;
    const size_t base = ((size_t)b * HW + (ok ? p : 0)) * C;
    const int slots = C >> 3;
    const float n0 = r0[(size_t)b * HW + (ok ? p : 0)], n1 = r1[(size_t)b * HW + (ok ? p : 0)];
    const float A = n0 + LPIPS_EPS, Bn = n1 + LPIPS_EPS;
    const float iA = 1.f / A, iB = 1.f / Bn;
    float dot0 = 0.f, dot1 = 0.f;
    for (int q = s; q < slots; q += 8) {
        float x[8], y[8];
        lp_unpack8(*reinterpret_cast<const uint4*>(f0 + base + q * 8), x);
        lp_unpack8(*reinterpret_cast<const uint4*>(f1 + base + q * 8), y);
        const float4 l0 = *reinterpret_cast<const float4*>(lin + q * 8), l1 = *reinterpret_cast<const float4*>(lin + q * 8 + 4);
        const float l[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float lt = l[e] * (x[e] * iA - y[e] * iB);
            dot0 = fmaf(lt, x[e], dot0);
            dot1 = fmaf(lt, y[e], dot1);
        }
    }
    dot0 = lp_octet_sum(dot0);
    dot1 = lp_octet_sum(dot1);
    if (!ok) return;
    const float gs = gout[b] / (float)HW;
    const float k0 = 2.f * dot0 / (A * A * n0), k1 = 2.f * dot1 / (Bn * Bn * n1);
    for (int q = s; q < slots; q += 8) {
        float x[8], y[8], a[8], c[8];
        lp_unpack8(*reinterpret_cast<const uint4*>(f0 + base + q * 8), x);
        lp_unpack8(*reinterpret_cast<const uint4*>(f1 + base + q * 8), y);
        const float4 l0 = *reinterpret_cast<const float4*>(lin + q * 8), l1 = *reinterpret_cast<const float4*>(lin + q * 8 + 4);
        const float l[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float lt2 = 2.f * l[e] * (x[e] * iA - y[e] * iB);
            a[e] = gs * (lt2 * iA - k0 * x[e]);
            c[e] = gs * (k1 * y[e] - lt2 * iB);
        }
        if (g0) *reinterpret_cast<uint4*>(g0 + base + q * 8) = make_uint4(lp_pack2(a[0], a[1]), lp_pack2(a[2], a[3]), lp_pack2(a[4], a[5]), lp_pack2(a[6], a[7]));
        if (g1) *reinterpret_cast<uint4*>(g1 + base + q * 8) = make_uint4(lp_pack2(c[0], c[1]), lp_pack2(c[2], c[3]), lp_pack2(c[4], c[5]), lp_pack2(c[6], c[7]));
    }
}


// ---- layout bridges between the library's fp32 NCHW tensors and this library's bf16 NHWC ones (round 6) ------------------------
// The frozen networks keep their forward (classifier) or their stem (LPIPS) on the library's fp32 NCHW convolutions while their
// gradient path / remaining layers run on the bf16 NHWC kernels.  ATen's strided copy did the conversion at 45 us per ResNet
// activation (16 per classifier call: 0.73 ms); a 64-channel x 64-pixel tile through LDS reads and writes whole lines.
//   fwd: y[b][p][c] = bf16( relu ? max(x[b][c][p], 0) : x[b][c][p] )
//   bwd: gx[b][c][p] = float(g[b][p][c]) * (gate == nullptr || gate[b][p][c] > 0)
constexpr int LB_T = 64;

__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int C,
                                                                    int HW, int relu) {
    __shared__ float tile[LB_T][LB_T + 1];
    const int b = blockIdx.z, c0 = blockIdx.y * LB_T, p0 = blockIdx.x * LB_T, tid = threadIdx.x;
    const float* xb = x + (size_t)b * C * HW;
#pragma unroll
    for (int i = 0; i < LB_T / 4; ++i) {
        const int c = i * 4 + (tid >> 6), p = tid & 63;
        float v = 0.f;
        if (c0 + c < C && p0 + p < HW) v = xb[(size_t)(c0 + c) * HW + p0 + p];
        tile[c][p] = relu ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pix = (tid >> 3) + 32 * k, slot = tid & 7;
        if (p0 + pix < HW && c0 + slot * 8 < C) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[slot * 8 + e][pix];
            *reinterpret_cast<uint4*>(y + ((size_t)b * HW + p0 + pix) * C + c0 + slot * 8) =
                make_uint4(lp_pack2(v[0], v[1]), lp_pack2(v[2], v[3]), lp_pack2(v[4], v[5]), lp_pack2(v[6], v[7]));
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const unsigned short* __restrict__ g,
                                                                    const unsigned short* __restrict__ gate, float* __restrict__ gx,
                                                                    int C, int HW) {
    __shared__ float tile[LB_T][LB_T + 1];
    const int b = blockIdx.z, c0 = blockIdx.y * LB_T, p0 = blockIdx.x * LB_T, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pix = (tid >> 3) + 32 * k, slot = tid & 7;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p0 + pix < HW && c0 + slot * 8 < C) {
            const size_t o = ((size_t)b * HW + p0 + pix) * C + c0 + slot * 8;
            lp_unpack8(*reinterpret_cast<const uint4*>(g + o), v);
            if (gate) {
                float q[8];
                lp_unpack8(*reinterpret_cast<const uint4*>(gate + o), q);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = q[e] > 0.f ? v[e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[slot * 8 + e][pix] = v[e];
    }
    __syncthreads();
    float* gb = gx + (size_t)b * C * HW;
#pragma unroll
    for (int i = 0; i < LB_T / 4; ++i) {
        const int c = i * 4 + (tid >> 6), p = tid & 63;
        if (c0 + c < C && p0 + p < HW) gb[(size_t)(c0 + c) * HW + p0 + p] = tile[c][p];
    }
}

// ---- classifier input: bilinear resize (align_corners = false, no antialias) + per-channel normalisation (round 6) --------------
// ResNet.classify_images (reference stylex/resnet_classifier.py:56-71): torchvision's tensor resize to 224 x 224 ==
// F.interpolate(mode='bilinear', align_corners=False), then (x - mean[c]) / std[c].  ATen's upsample kernel takes 140-160 us per
// call for the 32 x 3 x 224 x 224 output (and as long again backward), plus the sub / div passes and a layout copy of the generated
// batch in front; one pass each way here.  Index arithmetic as ATen's (area_pixel_compute_source_index): src = in / out * (dst + 0.5)
// - 0.5, clamped at 0; i0 = (int)src, i1 = i0 + (i0 < in - 1), lambda1 = src - i0.  The input is read through its four strides (the
// generator's channels_last output needs no dense copy).  Backward = the exact adjoint as a gather: an input pixel collects from
// the outputs whose i0 / i1 name it (fixed order: deterministic).
struct RsIdx { int i0, i1; float l0, l1; };
__device__ __forceinline__ RsIdx rs_index(int dst, float ratio, int in) {
    float src = ratio * (dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    RsIdx r;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

__global__ __launch_bounds__(256) void resize_norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ mean, const float* __restrict__ stdv, int C,
                                                              int Hi, int Wi, int Ho, int Wo, long sb, long sc, long sh, long sw,
                                                              unsigned total) {
    const float rh = (float)Hi / Ho, rw = (float)Wi / Wo;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int ox = i % Wo, oy = (i / Wo) % Ho, c = (i / (Wo * Ho)) % C, b = i / (Wo * Ho * C);
        const RsIdx h = rs_index(oy, rh, Hi), w = rs_index(ox, rw, Wi);
        const float* p = x + b * sb + c * sc;
        // the two rows as SCALAR fmas, one strictly after the other: hipcc's SLP vectoriser paired them into a packed fp32
        // fma with a crossed operand — the operand form whose low half went missing in torgb.hip when other waves shared
        // the GPU (tests/test_kernel_resources.py keeps every translation unit free of it)
        float p10 = p[h.i1 * sh + w.i0 * sw], p11 = p[h.i1 * sh + w.i1 * sw];
        float r0 = w.l0 * p[h.i0 * sh + w.i0 * sw] + w.l1 * p[h.i0 * sh + w.i1 * sw];
        asm volatile("" : "+v"(r0), "+v"(p10), "+v"(p11));
        const float r1 = w.l0 * p10 + w.l1 * p11;
        const float v = h.l0 * r0 + h.l1 * r1;
        y[i] = mean ? (v - mean[c]) / stdv[c] : v;
    }
}

__global__ __launch_bounds__(256) void resize_norm_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                              const float* __restrict__ stdv, int C, int Hi, int Wi, int Ho, int Wo,
                                                              unsigned total) {
    const float rh = (float)Hi / Ho, rw = (float)Wi / Wo;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int ix = i % Wi, iy = (i / Wi) % Hi, c = (i / (Wi * Hi)) % C, b = i / (Wi * Hi * C);
        // outputs whose source coordinate lies within (i - 1, i + 1), one more on either side for the float arithmetic
        int oy0 = (int)floorf((iy - 0.5f) / rh - 0.5f) - 1, oy1 = (int)ceilf((iy + 1.5f) / rh - 0.5f) + 1;
        int ox0 = (int)floorf((ix - 0.5f) / rw - 0.5f) - 1, ox1 = (int)ceilf((ix + 1.5f) / rw - 0.5f) + 1;
        oy0 = oy0 < 0 ? 0 : oy0;
        ox0 = ox0 < 0 ? 0 : ox0;
        oy1 = oy1 > Ho - 1 ? Ho - 1 : oy1;
        ox1 = ox1 > Wo - 1 ? Wo - 1 : ox1;
        const float* g = gy + ((size_t)b * C + c) * Ho * Wo;
        float acc = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            const RsIdx h = rs_index(oy, rh, Hi);
            const float wy = (h.i0 == iy ? h.l0 : 0.f) + (h.i1 == iy ? h.l1 : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const RsIdx w = rs_index(ox, rw, Wi);
                const float wx = (w.i0 == ix ? w.l0 : 0.f) + (w.i1 == ix ? w.l1 : 0.f);
                row = fmaf(wx, g[oy * Wo + ox], row);
            }
            acc = fmaf(wy, row, acc);
        }
        gx[i] = stdv ? acc / stdv[c] : acc;
    }
}

// out = (y > 0) ? a + b : 0 on bf16 NHWC tensors (b may be NULL): the ReLU gate of a BasicBlock's output applied to the sum of the
// two gradients that reach it (conv path + identity path) — one pass instead of an add and a gating pass (round 6)
__global__ __launch_bounds__(256) void relu_gate_add_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                            const uint4* __restrict__ y, uint4* __restrict__ out, unsigned total) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        float va[8], vb[8], vy[8];
        lp_unpack8(a[i], va);
        lp_unpack8(y[i], vy);
        if (b) {
            lp_unpack8(b[i], vb);
#pragma unroll
            for (int e = 0; e < 8; ++e) va[e] += vb[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) va[e] = vy[e] > 0.f ? va[e] : 0.f;
        out[i] = make_uint4(lp_pack2(va[0], va[1]), lp_pack2(va[2], va[3]), lp_pack2(va[4], va[5]), lp_pack2(va[6], va[7]));
    }
}

// ---- LPIPS-AlexNet's max-pools on the bf16 NHWC taps (round 6): 3 x 3 window, stride 2, no padding (torchvision alexnet
// features[2] / [5]; reference stylex_train.py:404 through lpips 0.1.4).  ATen's channels_last bf16 kernel took 213 us per call
// at B = 32 (16 MB in, 4 MB out: 75 GB/s) — 0.85 ms per train step for the four forward calls.  A lane owns 8 consecutive
// channels of one output pixel (nine 16-byte loads); the window position of the maximum is kept (one byte per element, ATen's
// rule: the FIRST maximum in (kh, kw) scan order, a NaN wins) so that the backward is a gather: an input pixel lies in at most
// 2 x 2 windows and adds, in ascending window order, the gradients of those that chose it (fp32 sum, one rounding).
__global__ __launch_bounds__(256) void maxpool3s2_nhwc_fwd_kernel(const uint4* __restrict__ x, uint4* __restrict__ y,
                                                                  uint2* __restrict__ idx, int Hi, int Wi, int Ho, int Wo, int C8,
                                                                  unsigned total) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cg = i % C8, ow = (i / C8) % Wo, oh = (i / (C8 * Wo)) % Ho, b = i / (C8 * Wo * Ho);
        const uint4* xb = x + ((long)b * Hi * Wi) * C8 + cg;
        float m[8];
        unsigned pos[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -__builtin_inff(), pos[e] = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float v[8];
                lp_unpack8(xb[((long)(2 * oh + kh) * Wi + (2 * ow + kw)) * C8], v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > m[e] || v[e] != v[e]) m[e] = v[e], pos[e] = kh * 3 + kw;
            }
        y[i] = make_uint4(lp_pack2(m[0], m[1]), lp_pack2(m[2], m[3]), lp_pack2(m[4], m[5]), lp_pack2(m[6], m[7]));
        idx[i] = make_uint2(pos[0] | pos[1] << 8 | pos[2] << 16 | pos[3] << 24, pos[4] | pos[5] << 8 | pos[6] << 16 | pos[7] << 24);
    }
}

__global__ __launch_bounds__(256) void maxpool3s2_nhwc_bwd_kernel(const uint4* __restrict__ gy, const uint2* __restrict__ idx,
                                                                  uint4* __restrict__ gx, int Hi, int Wi, int Ho, int Wo, int C8,
                                                                  unsigned total) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cg = i % C8, iw = (i / C8) % Wi, ih = (i / (C8 * Wi)) % Hi, b = i / (C8 * Wi * Hi);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        const int oh0 = ih >= 2 ? (int)(ih - 1) / 2 : 0, oh1 = min((int)ih / 2, Ho - 1);  // windows with 2*oh <= ih <= 2*oh + 2
        const int ow0 = iw >= 2 ? (int)(iw - 1) / 2 : 0, ow1 = min((int)iw / 2, Wo - 1);
        for (int oh = oh0; oh <= oh1; ++oh)
            for (int ow = ow0; ow <= ow1; ++ow) {
                const long o = (((long)b * Ho + oh) * Wo + ow) * C8 + cg;
                const unsigned me = (ih - 2 * oh) * 3 + (iw - 2 * ow);
                const uint2 p = idx[o];
                float g[8];
                lp_unpack8(gy[o], g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned pe = ((e < 4 ? p.x : p.y) >> (8 * (e & 3))) & 0xffu;
                    acc[e] += pe == me ? g[e] : 0.f;
                }
            }
        gx[i] = make_uint4(lp_pack2(acc[0], acc[1]), lp_pack2(acc[2], acc[3]), lp_pack2(acc[4], acc[5]), lp_pack2(acc[6], acc[7]));
    }
}

// ---- input gradient of a frozen network's FIRST convolution (round 6) -------------------------------------------------------
// dx[b][c][ih][iw] = sum_n sum_{kh,kw} dy[b][n][(ih + pad - kh) / S][(iw + pad - kw) / S] * w[n][c][kh][kw]   (exact divisions only)
// for the K x K / stride-S stems whose input is the 3-channel image: ResNet conv1 (7 x 7, stride 2, pad 3; torchvision
// resnet.py) and LPIPS-AlexNet's first layer (11 x 11, stride 4, pad 2; reference stylex_train.py:404, lpips 0.1.4) — the only
// gradient these networks owe the generator.  The libraries run it as a dense transposed convolution (0.67 - 0.75 ms per
// call at B = 32: a 3-column GEMM over K^2 * 64 with (1 - 1/S^2) of the taps structurally zero); here a thread owns one input
// pixel, a block one residue class ((ih + pad) % S, (iw + pad) % S), so the live taps and their weights are block-uniform
// (staged once per block in LDS, read as broadcasts) and consecutive lanes read consecutive dy elements, 8 independent loads in
// flight per lane; a thread owns FOUR vertically adjacent pixels of its class, so a dy row is loaded once for the up to four
// (pixel, tap) pairs it serves and an LDS weight read feeds up to four pixels (version 1, one pixel per thread, ran at the
// library's speed: 0.41 / 0.54 ms — its LDS broadcasts and loads, not its FMAs, were the cost).  fp32 FMA chain in a fixed
// (kw, dy row, n) order: deterministic.
constexpr int IG_PX = 4;  // vertically adjacent pixels (of one residue class) per thread

template <int S>
__global__ __launch_bounds__(256) void conv_image_grad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int N, int Ho, int Wo, int C, int K, int pad,
                                                              int Hi, int Wi, int tiles_w) {
    // blockIdx.x = (tile of the class's pixel grid: 16 rows x 64 columns), blockIdx.y = residue class rh * S + rw, blockIdx.z = b
    extern __shared__ __attribute__((aligned(16))) float wl[];  // the class's weights: [tap u][tap t][n][4] (channel padded to 4)
    const int rh = blockIdx.y / S, rw = blockIdx.y - rh * S;
    const int nth = (K - rh + S - 1) / S, ntw = (K - rw + S - 1) / S;  // taps of the class along each axis (block-uniform)
    for (int e = threadIdx.x; e < ntw * nth * N * 4; e += 256) {
        const int c = e & 3, n = (e >> 2) % N, tap = (e >> 2) / N;
        const int u = tap / nth, t = tap - u * nth;
        wl[e] = c < C ? w[((size_t)(n * C + c) * K + (rh + S * t)) * K + (rw + S * u)] : 0.f;
    }
    __syncthreads();
    const int ih0 = ((rh - pad) % S + S) % S, iw0 = ((rw - pad) % S + S) % S;  // first pixel of the class
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int i0 = (th * 4 + (threadIdx.x >> 6)) * IG_PX, j = tw * 64 + (threadIdx.x & 63);  // class coordinates of pixel 0
    const int iw = iw0 + S * j;
    const bool okw = iw < Wi;
    // kh = rh + S t  ->  oh = (ih + pad) / S - t; pixel p of the thread: qh = qh0 + p
    const int qh0 = (ih0 + S * i0 + pad) / S, qw = (iw + pad) / S;
    const int b = blockIdx.z;
    const size_t plane = (size_t)Ho * Wo;
    const float* dyb = dy + (size_t)b * N * plane;
    float acc[IG_PX][3];
#pragma unroll
    for (int p = 0; p < IG_PX; ++p) acc[p][0] = acc[p][1] = acc[p][2] = 0.f;
    float acc3[IG_PX] = {0.f, 0.f, 0.f, 0.f};  // fourth image channel (C == 4 only)
    constexpr int U = 8;  // independent loads in flight per lane
    // a dy row rr (oh = qh0 + IG_PX - 1 - rr... counted downwards from the top pixel's first tap) serves pixel p through tap
    // t = p + (IG_PX - 1 - rr')...: written out below with oh as the loop variable
    const int oh_lo = qh0 - (nth - 1), oh_hi = qh0 + IG_PX - 1;  // rows of dy any of the thread's pixels can touch
    for (int u = 0; u < ntw; ++u) {
        const int ow = qw - u;
        const bool livew = okw && ow >= 0 && ow < Wo;
        const float4* wu = reinterpret_cast<const float4*>(wl) + (size_t)u * nth * N;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const bool live = livew && oh >= 0 && oh < Ho;
            const float* src = dyb + (live ? (size_t)oh * Wo + ow : 0);
            for (int n0 = 0; n0 < N; n0 += U) {
                float g[U];
#pragma unroll
                for (int v = 0; v < U; ++v) g[v] = (live && n0 + v < N) ? src[(size_t)(n0 + v) * plane] : 0.f;
#pragma unroll
                for (int p = 0; p < IG_PX; ++p) {
                    const int t = qh0 + p - oh;  // the tap through which this dy row reaches pixel p (block- and wave-uniform)
                    if (t >= 0 && t < nth) {
                        const float4* wt = wu + (size_t)t * N;
#pragma unroll
                        for (int v = 0; v < U; ++v) {
                            const float4 wv = wt[n0 + v < N ? n0 + v : 0];  // one address per wave: an LDS broadcast
                            acc[p][0] = fmaf(g[v], wv.x, acc[p][0]);
                            acc[p][1] = fmaf(g[v], wv.y, acc[p][1]);
                            acc[p][2] = fmaf(g[v], wv.z, acc[p][2]);
                            if (C == 4) acc3[p] = fmaf(g[v], wv.w, acc3[p]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < IG_PX; ++p) {
        const int ih = ih0 + S * (i0 + p);
        if (okw && ih < Hi) {
            float* o = dx + ((size_t)b * C * Hi + ih) * Wi + iw;
            o[0] = acc[p][0];
            if (C > 1) o[(size_t)Hi * Wi] = acc[p][1];
            if (C > 2) o[(size_t)2 * Hi * Wi] = acc[p][2];
            if (C > 3) o[(size_t)3 * Hi * Wi] = acc3[p];
        }
    }
}

// Register-tiled version (round 6, fourth): per (tap column u, 4 output channels) a thread holds the NT + 3 dy rows its four pixels
// can touch and the NT tap weights of those channels in registers — 4 pixels x NT taps x 4 channels x 3 image channels = 144-192 FMAs
// per 24-28 global loads and 12-16 LDS broadcasts (version 3 re-read a weight from LDS for every 3 FMAs, which was its whole cost).
// NT = ceil(K / S) taps per axis at most (<= 4: 11 x 11 / 4 -> 3, 7 x 7 / 2 -> 4); a residue class with fewer taps has zero weights
// in the LDS image for the missing ones.
template <int S, int NT>
__global__ __launch_bounds__(256) void conv_image_grad_rt_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int N, int Ho, int Wo, int C, int K, int pad,
                                                                 int Hi, int Wi, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [tap u][tap t < NT][n (padded to 4)][4]
    const int rh = blockIdx.y / S, rw = blockIdx.y - rh * S;
    const int nth = (K - rh + S - 1) / S, ntw = (K - rw + S - 1) / S;
    const int N4 = (N + 3) & ~3;
    for (int e = threadIdx.x; e < ntw * NT * N4 * 4; e += 256) {
        const int c = e & 3, n = (e >> 2) % N4, tap = (e >> 2) / N4;
        const int u = tap / NT, t = tap - u * NT;
        wl[e] = (c < C && n < N && t < nth) ? w[((size_t)(n * C + c) * K + (rh + S * t)) * K + (rw + S * u)] : 0.f;
    }
    __syncthreads();
    const int ih0 = ((rh - pad) % S + S) % S, iw0 = ((rw - pad) % S + S) % S;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int i0 = (th * 4 + (threadIdx.x >> 6)) * IG_PX, j = tw * 64 + (threadIdx.x & 63);
    const int iw = iw0 + S * j;
    const bool okw = iw < Wi;
    const int qh0 = (ih0 + S * i0 + pad) / S, qw = (iw + pad) / S;
    const int b = blockIdx.z;
    const size_t plane = (size_t)Ho * Wo;
    const float* dyb = dy + (size_t)b * N * plane;
    constexpr int ROWS = NT + IG_PX - 1;
    float acc[IG_PX][4];
#pragma unroll
    for (int p = 0; p < IG_PX; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.f;
    // row r of the window is dy row oh = qh0 - (NT - 1) + r; it reaches pixel p through tap t = p + NT - 1 - r
    long roff[ROWS];
    bool rok[ROWS];
    for (int u = 0; u < ntw; ++u) {
        const int ow = qw - u;
        const bool livew = okw && ow >= 0 && ow < Wo;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int oh = qh0 - (NT - 1) + r;
            rok[r] = livew && oh >= 0 && oh < Ho;
            roff[r] = rok[r] ? (long)oh * Wo + ow : 0;
        }
        const float4* wu = reinterpret_cast<const float4*>(wl) + (size_t)u * NT * N4;
        for (int n0 = 0; n0 < N4; n0 += 4) {
            float g[ROWS][4];
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int v = 0; v < 4; ++v) g[r][v] = (rok[r] && n0 + v < N) ? dyb[(size_t)(n0 + v) * plane + roff[r]] : 0.f;
            float4 wv[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) wv[t][v] = wu[(size_t)t * N4 + n0 + v];  // one address per wave: LDS broadcasts
#pragma unroll
            for (int p = 0; p < IG_PX; ++p)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int r = p + NT - 1 - t;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        acc[p][0] = fmaf(g[r][v], wv[t][v].x, acc[p][0]);
                        acc[p][1] = fmaf(g[r][v], wv[t][v].y, acc[p][1]);
                        acc[p][2] = fmaf(g[r][v], wv[t][v].z, acc[p][2]);
                        acc[p][3] = fmaf(g[r][v], wv[t][v].w, acc[p][3]);
                    }
                }
        }
    }
#pragma unroll
    for (int p = 0; p < IG_PX; ++p) {
        const int ih = ih0 + S * (i0 + p);
        if (okw && ih < Hi) {
            float* o = dx + ((size_t)b * C * Hi + ih) * Wi + iw;
            o[0] = acc[p][0];
            if (C > 1) o[(size_t)Hi * Wi] = acc[p][1];
            if (C > 2) o[(size_t)2 * Hi * Wi] = acc[p][2];
            if (C > 3) o[(size_t)3 * Hi * Wi] = acc[p][3];
        }
    }
}

// 16 channel slices per block (1024 threads) when the launch has fewer than ~4 blocks of 4 waves per CU and enough channels
inline bool lp_wide(unsigned blocks, int64_t B, int64_t C) { return (int64_t)blocks * B < 1024 && C >= 128; }

inline unsigned grid_for(unsigned long n) {
    unsigned long b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int stylex_affine_act_nchw_fwd(const float* x, const float* scale, const float* shift, const float* residual, float* y, int64_t B,
                               int64_t C, int64_t HW, int relu, void* stream) {
    if (!x || !scale || !shift || !y || B < 1 || C < 1 || HW < 1 || B * C * HW > 0x7fffffffLL) return STYLEX_EINVAL;
    const unsigned long total = (unsigned long)(B * C * HW);
    const bool v4 = HW % 4 == 0 && !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                                      reinterpret_cast<uintptr_t>(residual)) & 15);
    if (v4)
        hipLaunchKernelGGL(affine_act_fwd_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                           residual, y, (unsigned)(total / 4), (unsigned)HW, (unsigned)C, relu);
    else
        hipLaunchKernelGGL(affine_act_fwd_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                           residual, y, (unsigned)total, (unsigned)HW, (unsigned)C, relu);
    return (int)hipGetLastError();
}

int stylex_affine_act_nchw_bwd(const float* gy, const float* y, const float* scale, float* gx, float* gres, int64_t B, int64_t C,
                               int64_t HW, int relu, void* stream) {
    if (!gy || !scale || !gx || (relu && !y) || B < 1 || C < 1 || HW < 1 || B * C * HW > 0x7fffffffLL) return STYLEX_EINVAL;
    const unsigned long total = (unsigned long)(B * C * HW);
    const bool v4 = HW % 4 == 0 && !((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y) |
                                      reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(gres)) & 15);
    if (v4)
        hipLaunchKernelGGL(affine_act_bwd_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, gy, y, scale, gx,
                           gres, (unsigned)(total / 4), (unsigned)HW, (unsigned)C, relu);
    else
        hipLaunchKernelGGL(affine_act_bwd_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, gy, y, scale, gx,
                           gres, (unsigned)total, (unsigned)HW, (unsigned)C, relu);
    return (int)hipGetLastError();
}

int stylex_affine_relu_maxpool_fwd(const float* x, const float* scale, const float* shift, float* y, unsigned char* idx, int64_t B,
                                   int64_t C, int64_t H, int64_t W, void* stream) {
    if (!x || !scale || !shift || !y || B < 1 || C < 1 || H < 1 || W < 1 || B * C * H * W > 0x7fffffffLL) return STYLEX_EINVAL;
    const int Ho = (int)((H - 1) / 2 + 1), Wo = (int)((W - 1) / 2 + 1);  // floor((H + 2 - 3) / 2) + 1
    const unsigned long total = (unsigned long)(B * C) * Ho * Wo;
    hipLaunchKernelGGL(affine_relu_maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y,
                       idx, (unsigned)total, (int)C, (int)H, (int)W, Ho, Wo);
    return (int)hipGetLastError();
}

int stylex_affine_relu_maxpool_bwd(const float* gy, const unsigned char* idx, const float* scale, float* gx, int64_t B, int64_t C,
                                   int64_t H, int64_t W, void* stream) {
    if (!gy || !idx || !scale || !gx || B < 1 || C < 1 || H < 1 || W < 1 || B * C * H * W > 0x7fffffffLL) return STYLEX_EINVAL;
    const int Ho = (int)((H - 1) / 2 + 1), Wo = (int)((W - 1) / 2 + 1);
    const unsigned long total = (unsigned long)(B * C * H * W);
    hipLaunchKernelGGL(affine_relu_maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, gy, idx, scale, gx,
                       (unsigned)total, (int)C, (int)H, (int)W, Ho, Wo);
    return (int)hipGetLastError();
}

int stylex_lpips_tap_fwd(const float* f0, const float* f1, const float* lin, float* partial, float* r0, float* r1, int64_t B,
                         int64_t C, int64_t HW, int64_t partial_stride, void* stream) {
    if (!f0 || !f1 || !lin || !partial || B < 1 || B > 65535 || C < 1 || HW < 1 || B * C * HW > 0x7fffffffLL) return STYLEX_EINVAL;
    const unsigned blocks = (unsigned)((HW + LP_PIX - 1) / LP_PIX);
    if (partial_stride < (int64_t)blocks) return STYLEX_EINVAL;
    if (lp_wide(blocks, B, C))
        hipLaunchKernelGGL(lpips_tap_fwd_kernel<16>, dim3(blocks, (unsigned)B), dim3(1024), 0, (hipStream_t)stream, f0, f1, lin, partial,
                           r0, r1, (int)C, (int)HW, (long)partial_stride);
    else
        hipLaunchKernelGGL(lpips_tap_fwd_kernel<4>, dim3(blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, f0, f1, lin, partial, r0,
                           r1, (int)C, (int)HW, (long)partial_stride);
    return (int)hipGetLastError();
}

int stylex_lpips_tap_bwd(const float* f0, const float* f1, const float* lin, const float* r0, const float* r1, const float* gout,
                         float* g0, float* g1, int64_t B, int64_t C, int64_t HW, void* stream) {
    if (!f0 || !f1 || !lin || !r0 || !r1 || !gout || (!g0 && !g1) || B < 1 || B > 65535 || C < 1 || HW < 1 ||
        B * C * HW > 0x7fffffffLL)
        return STYLEX_EINVAL;
    const unsigned blocks = (unsigned)((HW + LP_PIX - 1) / LP_PIX);
    if (lp_wide(blocks, B, C))
        hipLaunchKernelGGL(lpips_tap_bwd_kernel<16>, dim3(blocks, (unsigned)B), dim3(1024), 0, (hipStream_t)stream, f0, f1, lin, r0, r1, gout,
                           g0, g1, (int)C, (int)HW);
    else
        hipLaunchKernelGGL(lpips_tap_bwd_kernel<4>, dim3(blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, f0, f1, lin, r0, r1, gout,
                           g0, g1, (int)C, (int)HW);
    return (int)hipGetLastError();
}

int stylex_conv_image_grad(const float* dy, const float* w, float* dx, const int64_t* sh, void* stream) {
    // sh = {B, N, Ho, Wo, C, K, stride, pad, Hi, Wi}
    if (!dy || !w || !dx || !sh) return STYLEX_EINVAL;
    const int64_t B = sh[0], N = sh[1], Ho = sh[2], Wo = sh[3], C = sh[4], K = sh[5], S = sh[6], pad = sh[7], Hi = sh[8], Wi = sh[9];
    if (B < 1 || B > 65535 || N < 1 || C < 1 || C > 4 || K < 1 || K > 15 || pad < 0 || pad >= K || Hi < 1 || Wi < 1) return STYLEX_EINVAL;
    if (S != 1 && S != 2 && S != 4) return STYLEX_EINVAL;
    if (Ho != (Hi + 2 * pad - K) / S + 1 || Wo != (Wi + 2 * pad - K) / S + 1 || Ho < 1 || Wo < 1) return STYLEX_EINVAL;
    if (B * N * Ho * Wo > 0x7fffffffLL || B * C * Hi * Wi > 0x7fffffffLL) return STYLEX_EINVAL;
    const int ch = (int)((Hi + S - 1) / S), cw = (int)((Wi + S - 1) / S);  // pixels of a residue class along each axis (at most)
    const int tiles_h = (ch + 4 * IG_PX - 1) / (4 * IG_PX), tiles_w = (cw + 63) / 64;
    const dim3 grid((unsigned)(tiles_h * tiles_w), (unsigned)(S * S), (unsigned)B);
    const int64_t nt = (K + S - 1) / S, max_taps = nt * nt;
    const size_t smem = (size_t)(max_taps * ((N + 3) & ~3) * 4 * sizeof(float));  // the largest residue class's weights
    if (smem > 64 * 1024) return STYLEX_EINVAL;
    if (nt <= 4 && nt >= 2) {  // register-tiled kernel (the two frozen stems: 11 x 11 / 4 -> 3 taps per axis, 7 x 7 / 2 -> 4)
        const size_t smem_rt = (size_t)(nt * 4 * ((N + 3) & ~3) * 4 * sizeof(float));  // [<= nt tap columns][NT <= 4][N4][4]
#define STYLEX_IMG_GRAD_RT(SS, NTT)                                                                                              \
    hipLaunchKernelGGL((conv_image_grad_rt_kernel<SS, NTT>), grid, dim3(256), smem_rt, (hipStream_t)stream, dy, w, dx, (int)N, (int)Ho, \
                       (int)Wo, (int)C, (int)K, (int)pad, (int)Hi, (int)Wi, tiles_w)
        if (S == 4 && nt == 3) STYLEX_IMG_GRAD_RT(4, 3);
        else if (S == 4) STYLEX_IMG_GRAD_RT(4, 4);
        else if (S == 2 && nt == 4) STYLEX_IMG_GRAD_RT(2, 4);
        else if (S == 2) STYLEX_IMG_GRAD_RT(2, 3);
        else if (S == 1 && nt == 3) STYLEX_IMG_GRAD_RT(1, 3);
        else STYLEX_IMG_GRAD_RT(1, 4);
#undef STYLEX_IMG_GRAD_RT
        return (int)hipGetLastError();
    }
#define STYLEX_IMG_GRAD(SS)                                                                                                        \
    hipLaunchKernelGGL(conv_image_grad_kernel<SS>, grid, dim3(256), smem, (hipStream_t)stream, dy, w, dx, (int)N, (int)Ho, (int)Wo, \
                       (int)C, (int)K, (int)pad, (int)Hi, (int)Wi, tiles_w)
    if (S == 1) STYLEX_IMG_GRAD(1);
    else if (S == 2) STYLEX_IMG_GRAD(2);
    else STYLEX_IMG_GRAD(4);
#undef STYLEX_IMG_GRAD
    return (int)hipGetLastError();
}

int stylex_lpips_tap_nhwc_fwd(const void* f0, const void* f1, const float* lin, float* partial, float* r0, float* r1, int64_t B,
                              int64_t C, int64_t HW, int64_t partial_stride, void* stream) {
    if (!f0 || !f1 || !lin || !partial || B < 1 || B > 65535 || C < 8 || C % 8 || HW < 1 || B * C * HW > 0x7fffffffLL) return STYLEX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(f0) | reinterpret_cast<uintptr_t>(f1) | reinterpret_cast<uintptr_t>(lin)) & 15) return STYLEX_EINVAL;
    const unsigned blocks = (unsigned)((HW + LPN_PIX - 1) / LPN_PIX);
    if (partial_stride < (int64_t)blocks) return STYLEX_EINVAL;
    hipLaunchKernelGGL(lpips_tap_nhwc_fwd_kernel, dim3(blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)f0,
                       (const unsigned short*)f1, lin, partial, r0, r1, (int)C, (int)HW, (long)partial_stride);
    return (int)hipGetLastError();
}

int stylex_lpips_tap_nhwc_bwd(const void* f0, const void* f1, const float* lin, const float* r0, const float* r1, const float* gout,
                              void* g0, void* g1, int64_t B, int64_t C, int64_t HW, void* stream) {
    if (!f0 || !f1 || !lin || !r0 || !r1 || !gout || (!g0 && !g1) || B < 1 || B > 65535 || C < 8 || C % 8 || HW < 1 ||
        B * C * HW > 0x7fffffffLL)
        return STYLEX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(f0) | reinterpret_cast<uintptr_t>(f1) | reinterpret_cast<uintptr_t>(lin) |
         reinterpret_cast<uintptr_t>(g0) | reinterpret_cast<uintptr_t>(g1)) & 15)
        return STYLEX_EINVAL;
    hipLaunchKernelGGL(lpips_tap_nhwc_bwd_kernel, dim3((unsigned)((HW + LPN_PIX - 1) / LPN_PIX), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short*)f0, (const unsigned short*)f1, lin, r0, r1, gout, (unsigned short*)g0,
                       (unsigned short*)g1, (int)C, (int)HW);
    return (int)hipGetLastError();
}

int stylex_nchw_f32_to_nhwc_bf16(const float* x, void* y, int64_t B, int64_t C, int64_t HW, int relu, void* stream) {
    if (!x || !y || B < 1 || B > 65535 || C < 8 || C % 8 || HW < 1 || B * C * HW > 0x7fffffffLL) return STYLEX_EINVAL;
    if (reinterpret_cast<uintptr_t>(y) & 15) return STYLEX_EINVAL;
    const dim3 grid((unsigned)((HW + LB_T - 1) / LB_T), (unsigned)((C + LB_T - 1) / LB_T), (unsigned)B);
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, (int)C, (int)HW, relu);
    return (int)hipGetLastError();
}

int stylex_nhwc_bf16_to_nchw_f32(const void* g, const void* gate, float* gx, int64_t B, int64_t C, int64_t HW, void* stream) {
    if (!g || !gx || B < 1 || B > 65535 || C < 8 || C % 8 || HW < 1 || B * C * HW > 0x7fffffffLL) return STYLEX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gate)) & 15) return STYLEX_EINVAL;
    const dim3 grid((unsigned)((HW + LB_T - 1) / LB_T), (unsigned)((C + LB_T - 1) / LB_T), (unsigned)B);
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)g,
                       (const unsigned short*)gate, gx, (int)C, (int)HW);
    return (int)hipGetLastError();
}

int stylex_resize_norm_fwd(const float* x, float* y, const float* mean, const float* stdv, const int64_t* sh, const int64_t* strides,
                           void* stream) {
    // sh = {B, C, Hi, Wi, Ho, Wo}; strides = element strides of x (b, c, h, w); y dense NCHW
    if (!x || !y || !sh || !strides || (mean && !stdv) || (!mean && stdv)) return STYLEX_EINVAL;
    for (int k = 0; k < 6; ++k)
        if (sh[k] < 1) return STYLEX_EINVAL;
    const int64_t total = sh[0] * sh[1] * sh[4] * sh[5];
    if (total > 0x7fffffffLL || sh[0] * sh[1] * sh[2] * sh[3] > 0x7fffffffLL) return STYLEX_EINVAL;
    hipLaunchKernelGGL(resize_norm_fwd_kernel, dim3(grid_for((unsigned long)total)), dim3(256), 0, (hipStream_t)stream, x, y, mean, stdv,
                       (int)sh[1], (int)sh[2], (int)sh[3], (int)sh[4], (int)sh[5], (long)strides[0], (long)strides[1], (long)strides[2],
                       (long)strides[3], (unsigned)total);
    return (int)hipGetLastError();
}

int stylex_resize_norm_bwd(const float* gy, float* gx, const float* stdv, const int64_t* sh, void* stream) {
    // gy dense [B][C][Ho][Wo] -> gx dense [B][C][Hi][Wi]; stdv NULL = no normalisation
    if (!gy || !gx || !sh) return STYLEX_EINVAL;
    for (int k = 0; k < 6; ++k)
        if (sh[k] < 1) return STYLEX_EINVAL;
    const int64_t total = sh[0] * sh[1] * sh[2] * sh[3];
    if (total > 0x7fffffffLL || sh[0] * sh[1] * sh[4] * sh[5] > 0x7fffffffLL) return STYLEX_EINVAL;
    hipLaunchKernelGGL(resize_norm_bwd_kernel, dim3(grid_for((unsigned long)total)), dim3(256), 0, (hipStream_t)stream, gy, gx, stdv,
                       (int)sh[1], (int)sh[2], (int)sh[3], (int)sh[4], (int)sh[5], (unsigned)total);
    return (int)hipGetLastError();
}

// sh = {B, Hi, Wi, C}: x / gx [B][Hi][Wi][C] bf16, y / gy [B][Ho][Wo][C] bf16, idx [B][Ho][Wo][C] bytes; Ho = (Hi - 3) / 2 + 1
static int maxpool_dims(const int64_t* sh, int* Ho, int* Wo) {
    if (!sh || sh[0] <= 0 || sh[1] < 3 || sh[2] < 3 || sh[3] < 8 || sh[3] % 8) return STYLEX_EINVAL;
    *Ho = (int)((sh[1] - 3) / 2 + 1);
    *Wo = (int)((sh[2] - 3) / 2 + 1);
    if (sh[0] * sh[1] * sh[2] * (sh[3] / 8) > 0x7fffffffLL) return STYLEX_EINVAL;
    return 0;
}

int stylex_maxpool3s2_nhwc_fwd(const void* x, void* y, void* idx, const int64_t* sh, void* stream) {
    int Ho, Wo;
    if (!x || !y || !idx || maxpool_dims(sh, &Ho, &Wo)) return STYLEX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15 || (reinterpret_cast<uintptr_t>(idx) & 7)) return STYLEX_EINVAL;
    const int C8 = (int)(sh[3] / 8);
    const unsigned total = (unsigned)(sh[0] * Ho * Wo * C8);
    hipLaunchKernelGGL(maxpool3s2_nhwc_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y,
                       (uint2*)idx, (int)sh[1], (int)sh[2], Ho, Wo, C8, total);
    return (int)hipGetLastError();
}

int stylex_maxpool3s2_nhwc_bwd(const void* gy, const void* idx, void* gx, const int64_t* sh, void* stream) {
    int Ho, Wo;
    if (!gy || !gx || !idx || maxpool_dims(sh, &Ho, &Wo)) return STYLEX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15 || (reinterpret_cast<uintptr_t>(idx) & 7)) return STYLEX_EINVAL;
    const int C8 = (int)(sh[3] / 8);
    const unsigned total = (unsigned)(sh[0] * sh[1] * sh[2] * C8);
    hipLaunchKernelGGL(maxpool3s2_nhwc_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)gy,
                       (const uint2*)idx, (uint4*)gx, (int)sh[1], (int)sh[2], Ho, Wo, C8, total);
    return (int)hipGetLastError();
}

int stylex_relu_gate_add(const void* a, const void* b, const void* y, void* out, int64_t numel, void* stream) {
    if (!a || !y || !out || numel < 8 || numel % 8 || numel > 0x7fffffffLL * 8) return STYLEX_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15)
        return STYLEX_EINVAL;
    const unsigned total = (unsigned)(numel / 8);
    hipLaunchKernelGGL(relu_gate_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b,
                       (const uint4*)y, (uint4*)out, total);
    return (int)hipGetLastError();
}

}  // extern "C"
