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

// Sum over the block's 4 waves (channel slices) of a per-lane value, in fixed order; every lane of every wave gets it.
__device__ __forceinline__ float lp_cross_wave(float v, float (*buf)[LP_PIX], int wave, int lane) {
    buf[wave][lane] = v;
    __syncthreads();
    const float r = (buf[0][lane] + buf[1][lane]) + (buf[2][lane] + buf[3][lane]);
    __syncthreads();
    return r;
}

// grid (ceil(HW / 64), B), block 256: lane = pixel (coalesced along HW), wave w takes channels w, w + 4, ...; the channel
// loops are unrolled 8x so that 16 independent loads are in flight per lane (a one-lane-per-pixel loop over 384 channels
// was pure load latency: 360 us per tap at 15x15).
__global__ __launch_bounds__(256) void lpips_tap_fwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                            const float* __restrict__ lin, float* __restrict__ partial,
                                                            float* __restrict__ r0, float* __restrict__ r1, int C, int HW,
                                                            long partial_stride) {
    __shared__ float buf[4][LP_PIX];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * LP_PIX + lane;
    const bool ok = p < HW;
    const float* a = f0 + (size_t)b * C * HW + (ok ? p : 0);
    const float* q = f1 + (size_t)b * C * HW + (ok ? p : 0);
    float s0 = 0.f, s1 = 0.f;
    int c = wave;
    for (; c + 28 < C; c += 32) {
        float x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + 4 * u) * HW];
            y[u] = q[(size_t)(c + 4 * u) * HW];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s0 = fmaf(x[u], x[u], s0);
            s1 = fmaf(y[u], y[u], s1);
        }
    }
    for (; c < C; c += 4) {
        const float x = a[(size_t)c * HW], y = q[(size_t)c * HW];
        s0 = fmaf(x, x, s0);
        s1 = fmaf(y, y, s1);
    }
    s0 = lp_cross_wave(s0, buf, wave, lane);
    s1 = lp_cross_wave(s1, buf, wave, lane);
    const float n0 = sqrtf(s0), n1 = sqrtf(s1);
    if (ok && wave == 0) {
        if (r0) r0[(size_t)b * HW + p] = n0;
        if (r1) r1[(size_t)b * HW + p] = n1;
    }
    const float i0 = 1.f / (n0 + LPIPS_EPS), i1 = 1.f / (n1 + LPIPS_EPS);
    float d = 0.f;
    c = wave;
    for (; c + 28 < C; c += 32) {
        float x[8], y[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + 4 * u) * HW];
            y[u] = q[(size_t)(c + 4 * u) * HW];
            l[u] = lin[c + 4 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float t = x[u] * i0 - y[u] * i1;
            d = fmaf(l[u] * t, t, d);
        }
    }
    for (; c < C; c += 4) {
        const float t = a[(size_t)c * HW] * i0 - q[(size_t)c * HW] * i1;
        d = fmaf(lin[c] * t, t, d);
    }
    d = lp_cross_wave(ok ? d : 0.f, buf, wave, lane);
    if (wave == 0) {  // fixed-order sum over the block's 64 pixels
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (lane == 0) partial[(size_t)b * partial_stride + blockIdx.x] = d / (float)HW;
    }
}

// d out[b] / d f1[c][p] = g[b] / HW * ( q_c / B - (sum_k q_k y_k) / (B^2 r1) * y_c ),  q_c = -2 lin_c (x_c / A - y_c / B),
// A = r0 + eps, B = r1 + eps  (and symmetrically for f0).  A pixel whose features are all zero gives 0 / 0 = NaN, as the
// reference's sqrt backward does.  Same thread layout as the forward.
__global__ __launch_bounds__(256) void lpips_tap_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                            const float* __restrict__ lin, const float* __restrict__ r0,
                                                            const float* __restrict__ r1, const float* __restrict__ gout,
                                                            float* __restrict__ g0, float* __restrict__ g1, int C, int HW) {
    __shared__ float buf[4][LP_PIX];
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
    for (; c + 28 < C; c += 32) {
        float x[8], y[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + 4 * u) * HW];
            y[u] = q[(size_t)(c + 4 * u) * HW];
            l[u] = lin[c + 4 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float lt = l[u] * (x[u] * iA - y[u] * iB);
            dot0 = fmaf(lt, x[u], dot0);
            dot1 = fmaf(lt, y[u], dot1);
        }
    }
    for (; c < C; c += 4) {
        const float x = a[(size_t)c * HW], y = q[(size_t)c * HW];
        const float lt = lin[c] * (x * iA - y * iB);
        dot0 = fmaf(lt, x, dot0);
        dot1 = fmaf(lt, y, dot1);
    }
    dot0 = lp_cross_wave(dot0, buf, wave, lane);
    dot1 = lp_cross_wave(dot1, buf, wave, lane);
    if (!ok) return;
    const float gs = gout[b] / (float)HW;
    // d/dx_c = 2 lin_c t_c / A - 2 dot0 / (A^2 n0) x_c ;  d/dy_c = -2 lin_c t_c / B + 2 dot1 / (B^2 n1) y_c
    const float k0 = 2.f * dot0 / (A * A * n0), k1 = 2.f * dot1 / (Bn * Bn * n1);
    c = wave;
    for (; c + 28 < C; c += 32) {
        float x[8], y[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = a[(size_t)(c + 4 * u) * HW];
            y[u] = q[(size_t)(c + 4 * u) * HW];
            l[u] = lin[c + 4 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float lt2 = 2.f * l[u] * (x[u] * iA - y[u] * iB);
            if (g0) g0[base + (size_t)(c + 4 * u) * HW] = gs * (lt2 * iA - k0 * x[u]);
            if (g1) g1[base + (size_t)(c + 4 * u) * HW] = gs * (k1 * y[u] - lt2 * iB);
        }
    }
    for (; c < C; c += 4) {
        const float x = a[(size_t)c * HW], y = q[(size_t)c * HW];
        const float lt2 = 2.f * lin[c] * (x * iA - y * iB);
        if (g0) g0[base + (size_t)c * HW] = gs * (lt2 * iA - k0 * x);
        if (g1) g1[base + (size_t)c * HW] = gs * (k1 * y - lt2 * iB);
    }
}

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
    hipLaunchKernelGGL(lpips_tap_fwd_kernel, dim3(blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, f0, f1, lin, partial, r0,
                       r1, (int)C, (int)HW, (long)partial_stride);
    return (int)hipGetLastError();
}

int stylex_lpips_tap_bwd(const float* f0, const float* f1, const float* lin, const float* r0, const float* r1, const float* gout,
                         float* g0, float* g1, int64_t B, int64_t C, int64_t HW, void* stream) {
    if (!f0 || !f1 || !lin || !r0 || !r1 || !gout || (!g0 && !g1) || B < 1 || B > 65535 || C < 1 || HW < 1 ||
        B * C * HW > 0x7fffffffLL)
        return STYLEX_EINVAL;
    hipLaunchKernelGGL(lpips_tap_bwd_kernel, dim3((unsigned)((HW + LP_PIX - 1) / LP_PIX), (unsigned)B), dim3(256), 0, (hipStream_t)stream, f0,
                       f1, lin, r0, r1, gout, g0, g1, (int)C, (int)HW);
    return (int)hipGetLastError();
}

}  // extern "C"
