// stylex_capi.hip — extern "C" boundary of libstylex_hip.so (declared in include/stylex_hip.h).
// Argument checking, GEMM-view parameter construction, launch, optional hipEvent timing.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <array>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "stylex_internal.h"

namespace {

typedef std::array<int64_t, 9> LayerKey;  // cls, B, Hi, Wi, C, N, KH, stride, s2d_c
struct TimedLaunch {
    hipEvent_t start, stop;
    int cls;
    double flops;
    double bytes;  // algorithmic HBM bytes of the launch: operands in + result out, each once
    LayerKey key;
    char kernel[112];  // rocprof name of the kernel the call launched (stylex_note_kernel), "" when none was noted
};
struct LayerAcc {
    int64_t launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};
std::map<LayerKey, LayerAcc> g_layers;
std::map<std::pair<int, std::string>, LayerAcc> g_kernels;  // (class, kernel name) -> totals
thread_local char t_kernel[112] = "";
std::mutex g_mu;
bool g_timing = false;
thread_local int t_timing_paused = 0;  // stylex_timing_pause: launches of the calling thread are not recorded (the frozen networks)
std::vector<TimedLaunch> g_pending;
int64_t g_launches[3] = {0, 0, 0};
double g_ms[3] = {0, 0, 0};
double g_flops[3] = {0, 0, 0};
double g_bytes[3] = {0, 0, 0};

struct ScopedTimer {
    bool on;
    TimedLaunch t;
    hipStream_t s;
    ScopedTimer(int cls, double flops, double bytes, hipStream_t stream, const int64_t* sh = nullptr, int s2d = 0)
        : on(g_timing && t_timing_paused == 0), s(stream) {
        if (!on) return;
        t.cls = cls;
        t.flops = flops;
        t.bytes = bytes;
        t.key = LayerKey{cls, 0, 0, 0, 0, 0, 0, 0, 0};
        if (sh) t.key = LayerKey{cls, sh[0], sh[1], sh[2], sh[3], sh[4], sh[5], sh[7], s2d};
        t_kernel[0] = 0;
        (void)hipEventCreate(&t.start);
        (void)hipEventCreate(&t.stop);
        (void)hipEventRecord(t.start, s);
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(t.stop, s);
        memcpy(t.kernel, t_kernel, sizeof(t.kernel));
        std::lock_guard<std::mutex> lk(g_mu);
        g_pending.push_back(t);
    }
};

void drain_pending() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& t : g_pending) {
        float ms = 0.f;
        (void)hipEventSynchronize(t.stop);
        (void)hipEventElapsedTime(&ms, t.start, t.stop);
        g_launches[t.cls] += 1;
        g_ms[t.cls] += ms;
        g_flops[t.cls] += t.flops;
        g_bytes[t.cls] += t.bytes;
        LayerAcc& la = g_layers[t.key];
        la.launches += 1;
        la.ms += ms;
        la.flops += t.flops;
        la.bytes += t.bytes;
        LayerAcc& ka = g_kernels[std::make_pair(t.cls, std::string(t.kernel))];
        ka.launches += 1;
        ka.ms += ms;
        ka.flops += t.flops;
        ka.bytes += t.bytes;
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    g_pending.clear();
}

// algorithmic bytes of one conv launch: activation tensor in + out (each once, at the storage width of the
// precision mode) + the weights once (fp32 master copy for the gradient, packed operand otherwise)
double conv_bytes(const int64_t* sh, int precision, bool wgrad) {
    const double es = precision == STYLEX_BF16_ACT ? 2.0 : 4.0;
    const double in = (double)sh[0] * sh[1] * sh[2] * sh[3], out = (double)sh[0] * sh[9] * sh[10] * sh[4];
    const double w = (double)sh[4] * sh[3] * sh[5] * sh[6];
    return (in + out) * es + w * (wgrad ? 4.0 : (precision == STYLEX_F32 ? 4.0 : 2.0));
}

bool conv_shape_ok(const int64_t* sh) {
    for (int i = 0; i < STYLEX_CONV_NSHAPE; ++i)
        if (sh[i] < 0 || sh[i] > 0x7fffffff) return false;
    int64_t B = sh[0], Hi = sh[1], Wi = sh[2], C = sh[3], N = sh[4], KH = sh[5], KW = sh[6], st = sh[7], pad = sh[8],
            Ho = sh[9], Wo = sh[10];
    if (B < 1 || Hi < 1 || Wi < 1 || C < 1 || N < 1) return false;
    // 1x1 and 3x3 are the StylEx layers; 5x5 (stride 1) is LPIPS-AlexNet's second layer (round 6: the generic implicit-GEMM
    // kernel gathers any K x K window; every specialised kernel checks its own 3x3 / 1x1 premise)
    // ... and 11x11 / stride 4 is its first layer (FORWARD only: the generic kernel's transposed gather knows strides 1 and 2;
    // stylex_conv_image_grad is that layer's data gradient)
    if (!((KH == 1 && KW == 1) || (KH == 3 && KW == 3) || (KH == 5 && KW == 5 && st == 1) || (KH == 11 && KW == 11 && st == 4))) return false;
    if (st != 1 && st != 2 && !(st == 4 && KH == 11)) return false;
    if (Ho != (Hi + 2 * pad - KH) / st + 1 || Wo != (Wi + 2 * pad - KW) / st + 1) return false;
    if (B * Ho * Wo > 0x7fffffff || B * Hi * Wi > 0x7fffffff) return false;
    return true;
}

void fill_common(ConvKParams& p, const int64_t* sh) {
    memset(&p, 0, sizeof(p));
    p.B = (int)sh[0];
    p.KH = (int)sh[5];
    p.KW = (int)sh[6];
    p.stride = (int)sh[7];
    p.pad = (int)sh[8];
}

}  // namespace

void stylex_note_kernel(const char* fmt, ...) {
    if (!g_timing) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_kernel, sizeof(t_kernel), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* stylex_version(void) { return "stylex-hip 0.1 (gfx950)"; }

int stylex_init(int device) {
    hipError_t e = hipSetDevice(device);
    return (int)e;
}

int stylex_timing_pause(int on) {
    t_timing_paused += on ? 1 : -1;
    if (t_timing_paused < 0) t_timing_paused = 0;
    return 0;
}

int stylex_timing_enable(int on) {
    drain_pending();
    std::lock_guard<std::mutex> lk(g_mu);
    g_timing = on != 0;
    if (on) {
        for (int i = 0; i < 3; ++i) {
            g_launches[i] = 0;
            g_ms[i] = 0;
            g_flops[i] = 0;
            g_bytes[i] = 0;
        }
        g_layers.clear();
        g_kernels.clear();
    }
    return 0;
}

// Per-KERNEL view of the same measurements: one row per (class, kernel name as rocprofv3 prints it).  names = cap
// slots of 112 bytes (NUL-terminated), meta[r] = {cls, launches}, vals[r] = {total ms, total algorithmic FLOPs, total
// algorithmic bytes}.  Returns the number of rows written (<= cap).
int stylex_timing_kernels(char* names, int64_t* meta, double* vals, int64_t cap) {
    if (!names || !meta || !vals) return STYLEX_EINVAL;
    drain_pending();
    std::lock_guard<std::mutex> lk(g_mu);
    int64_t r = 0;
    for (const auto& kv : g_kernels) {
        if (r >= cap) break;
        snprintf(names + r * 112, 112, "%s", kv.first.second.c_str());
        meta[r * 2 + 0] = kv.first.first;
        meta[r * 2 + 1] = kv.second.launches;
        vals[r * 3 + 0] = kv.second.ms;
        vals[r * 3 + 1] = kv.second.flops;
        vals[r * 3 + 2] = kv.second.bytes;
        ++r;
    }
    return (int)r;
}

// Per-layer view of the same measurements: one row per (class, conv shape).  meta[r] = {cls, B, Hi, Wi, C, N, KH,
// stride, s2d_c, launches}, vals[r] = {total ms, total algorithmic FLOPs, total algorithmic bytes}.  Returns the
// number of rows written (<= cap).
int stylex_timing_layers(int64_t* meta, double* vals, int64_t cap) {
    drain_pending();
    std::lock_guard<std::mutex> lk(g_mu);
    int64_t r = 0;
    for (const auto& kv : g_layers) {
        if (r >= cap) break;
        for (int i = 0; i < 9; ++i) meta[r * 10 + i] = kv.first[i];
        meta[r * 10 + 9] = kv.second.launches;
        vals[r * 3 + 0] = kv.second.ms;
        vals[r * 3 + 1] = kv.second.flops;
        vals[r * 3 + 2] = kv.second.bytes;
        ++r;
    }
    return (int)r;
}

int stylex_timing_report(int cls, int64_t* launches, double* total_ms, double* total_flops, double* total_bytes) {
    if (cls < 0 || cls > 2) return STYLEX_EINVAL;
    drain_pending();
    std::lock_guard<std::mutex> lk(g_mu);
    if (launches) *launches = g_launches[cls];
    if (total_ms) *total_ms = g_ms[cls];
    if (total_flops) *total_flops = g_flops[cls];
    if (total_bytes) *total_bytes = g_bytes[cls];
    return 0;
}

int stylex_pack_weight(const float* w, void* wf, void* wb, const int64_t* sh, int precision, void* stream) {
    if (!w || sh[0] < 1 || sh[1] < 1 || sh[2] < 1 || sh[3] < 1) return STYLEX_EINVAL;
    if (precision != STYLEX_F32 && precision != STYLEX_BF16 && precision != STYLEX_BF16_ACT) return STYLEX_EINVAL;
    return stylex_launch_pack(w, wf, wb, (int)sh[0], (int)sh[1], (int)(sh[2] * sh[3]),
                              precision == STYLEX_F32 ? STYLEX_F32 : STYLEX_BF16, (hipStream_t)stream);
}

static void fwd_params(ConvKParams& p, const int64_t* sh) {
    fill_common(p, sh);
    p.Hi = (int)sh[1];
    p.Wi = (int)sh[2];
    p.Ck = (int)sh[3];
    p.N = (int)sh[4];
    p.Ho = (int)sh[9];
    p.Wo = (int)sh[10];
    p.M = p.B * p.Ho * p.Wo;
}

static void bwd_data_params(ConvKParams& p, const int64_t* sh) {
    fill_common(p, sh);
    // source = dy [B][Ho_f][Wo_f][N_f];  dest = dx [B][Hi_f][Wi_f][C_f]
    p.Hi = (int)sh[9];
    p.Wi = (int)sh[10];
    p.Ck = (int)sh[4];
    p.N = (int)sh[3];
    p.Ho = (int)sh[1];
    p.Wo = (int)sh[2];
    p.M = p.B * p.Ho * p.Wo;
}

int64_t stylex_conv2d_workspace_bytes(const int64_t* sh, int which, int precision) {
    if (!conv_shape_ok(sh) || (which != 0 && which != 1)) return STYLEX_EINVAL;
    ConvKParams p;
    if (which == 0) fwd_params(p, sh);
    else {
        bwd_data_params(p, sh);
        p.transposed = 1;
    }
    p.act_bf16 = precision == STYLEX_BF16_ACT;
    return stylex_igemm_workspace_bytes(p, precision);
}

int stylex_conv_mask_supported(const int64_t* sh, int which, int flags, int precision) {
    if (!conv_shape_ok(sh) || precision != STYLEX_BF16_ACT || (which != 0 && which != 1)) return 0;
    // the same applicability checks the launchers run, on 16-byte aligned stand-in pointers, launching nothing
    static const uintptr_t fake = 4096;
    ConvKParams p;
    if (which == 0) {
        if (!(flags & STYLEX_EPI_MASK_OUT) || !(flags & STYLEX_EPI_LRELU)) return 0;
        fwd_params(p, sh);
        if (p.N % 8 != 0) return 0;
        p.mask = (unsigned char*)fake;
    } else {
        if (!(flags & STYLEX_EPI_GATE_MASK) || (flags & ~(STYLEX_EPI_GATE_MASK))) return 0;
        bwd_data_params(p, sh);
        if (p.KH != 3 || p.stride != 1 || p.pad != 1) return 0;
        p.flip_taps = 1;
        p.gate_mask = (const unsigned char*)fake;
    }
    p.a = (const float*)fake;
    p.w = (const void*)fake;
    p.y = (float*)fake;
    p.bias = (const float*)fake;
    p.flags = flags;
    p.act_bf16 = 1;
    p.dry = 1;
    if (which == 0 && stylex_launch_rgb(p, nullptr) == 0) return 1;
    return stylex_launch_halo(p, nullptr) == 0 ? 1 : 0;
}

int stylex_conv2d_fwd(const void* x, const void* w_fwd, void* y, const int64_t* sh, int flags,
                      const stylex_conv_epilogue* epi, int precision, void* workspace, int64_t workspace_bytes,
                      void* stream) {
    if (!x || !w_fwd || !y || !conv_shape_ok(sh)) return STYLEX_EINVAL;
    if (precision != STYLEX_F32 && precision != STYLEX_BF16 && precision != STYLEX_BF16_ACT) return STYLEX_EINVAL;
    ConvKParams p;
    fwd_params(p, sh);
    p.a = (const float*)x;
    p.w = w_fwd;
    p.y = (float*)y;
    p.flags = flags;
    p.act_bf16 = precision == STYLEX_BF16_ACT;
    if (p.act_bf16) precision = STYLEX_BF16;
    if (epi) {
        p.a_scale = epi->in_scale;
        p.bias = epi->bias;
        p.out_scale = epi->out_scale;
        p.noise = epi->noise;
        p.noise_stride = (int)epi->noise_stride;
        p.noise_w = epi->noise_w;
        p.noise_b = epi->noise_b;
        p.residual = (const float*)epi->residual;
        p.res_scale = epi->res_scale;
        p.s2d_c = (int)epi->s2d_c;
    }
    if (flags & STYLEX_EPI_GATE_MASK) return STYLEX_EINVAL;  // data gradient only
    if (flags & STYLEX_EPI_MASK_OUT) {
        if (!epi || !epi->mask || !(flags & STYLEX_EPI_LRELU) || !p.act_bf16 || p.N % 8 != 0) return STYLEX_EINVAL;
        p.mask = (unsigned char*)epi->mask;
    }
    if ((flags & STYLEX_EPI_BIAS) && !p.bias) return STYLEX_EINVAL;
    if ((flags & STYLEX_EPI_OSCALE) && !p.out_scale) return STYLEX_EINVAL;
    if ((flags & STYLEX_EPI_NOISE_NAT) && (!(flags & STYLEX_EPI_NOISE) || (reinterpret_cast<uintptr_t>(p.noise) & 15) ||
                                           p.noise_stride % 4 != 0))
        return STYLEX_EINVAL;
    if ((flags & STYLEX_EPI_NOISE) && (!p.noise || !p.noise_w || !p.noise_b || p.noise_stride < p.Ho || p.noise_stride < p.Wo))
        return STYLEX_EINVAL;
    if ((flags & STYLEX_EPI_RESIDUAL) && !p.residual) return STYLEX_EINVAL;
    if (p.s2d_c && (p.Ck != 4 * p.s2d_c || p.KH != 3 || p.stride != 1 || p.pad != 1 || p.s2d_c % 64)) return STYLEX_EINVAL;
    double flops = 2.0 * p.M * (double)p.N * p.Ck * p.KH * p.KW / (p.s2d_c ? 4.0 : 1.0);  // algorithmic: 9*C, not 36*C
    ScopedTimer tm(0, flops, conv_bytes(sh, p.act_bf16 ? STYLEX_BF16_ACT : precision, false), (hipStream_t)stream, sh, p.s2d_c);
    return stylex_launch_igemm(p, precision, workspace, workspace_bytes, (hipStream_t)stream);
}

static int s2d_res_params(ConvKParams& p, const int64_t* sh, int64_t s2d_c, int64_t res_c) {
    if (!conv_shape_ok(sh)) return STYLEX_EINVAL;
    fwd_params(p, sh);
    p.s2d_c = (int)s2d_c;
    p.c2 = (int)res_c;
    if (s2d_c <= 0 || s2d_c % 64 || p.Ck != 4 * p.s2d_c || p.KH != 3 || p.stride != 1 || p.pad != 1) return STYLEX_EINVAL;
    if (res_c < 8 || res_c % 8) return STYLEX_EINVAL;
    p.act_bf16 = 1;
    p.flags = STYLEX_EPI_BIAS;
    return 0;
}

int stylex_conv2d_s2d_res_supported(const int64_t* sh, int64_t s2d_c, int64_t res_c) {
    static const uintptr_t fake = 4096;
    ConvKParams p;
    if (s2d_res_params(p, sh, s2d_c, res_c)) return 0;
    p.a = (const float*)fake;
    p.w = (const void*)fake;
    p.y = (float*)fake;
    p.bias = (const float*)fake;
    p.x2 = p.w2 = (const void*)fake;
    p.dry = 1;
    return stylex_launch_halo_dma(p, nullptr) == 0 ? 1 : 0;
}

int stylex_conv2d_s2d_res_fwd(const void* x_s2d, const void* w_fwd_s2d, const void* x_res, const void* w_res, const float* bias,
                              void* y, const int64_t* sh, int64_t s2d_c, int64_t res_c, float scale, void* stream) {
    if (!x_s2d || !w_fwd_s2d || !x_res || !w_res || !bias || !y) return STYLEX_EINVAL;
    ConvKParams p;
    if (int rc = s2d_res_params(p, sh, s2d_c, res_c)) return rc;
    p.a = (const float*)x_s2d;
    p.w = w_fwd_s2d;
    p.y = (float*)y;
    p.bias = bias;
    p.x2 = x_res;
    p.w2 = w_res;
    p.res_scale = scale;
    const double flops = 2.0 * p.M * (double)p.N * (p.Ck * 9 / 4.0 + p.c2);
    ScopedTimer tm(0, flops, conv_bytes(sh, STYLEX_BF16_ACT, false) + 2.0 * p.M * p.c2, (hipStream_t)stream, sh, p.s2d_c);
    const int rc = stylex_launch_halo_dma(p, (hipStream_t)stream);
    return rc == STYLEX_NOT_APPLICABLE ? STYLEX_EINVAL : rc;
}

int stylex_conv2d_bwd_data(const void* dy, const void* w_bwd, void* dx, const int64_t* sh, int flags,
                           const stylex_conv_epilogue* epi, int precision, void* workspace, int64_t workspace_bytes,
                           void* stream) {
    if (!dy || !w_bwd || !dx || !conv_shape_ok(sh) || sh[7] > 2) return STYLEX_EINVAL;
    if (precision != STYLEX_F32 && precision != STYLEX_BF16 && precision != STYLEX_BF16_ACT) return STYLEX_EINVAL;
    if (flags & ~(STYLEX_EPI_OSCALE | STYLEX_EPI_GATE | STYLEX_EPI_GATE_MASK)) return STYLEX_EINVAL;
    if ((flags & STYLEX_EPI_GATE) && (!epi || !epi->residual)) return STYLEX_EINVAL;
    if ((flags & STYLEX_EPI_GATE_MASK) && ((flags & STYLEX_EPI_GATE) || !epi || !epi->mask || precision != STYLEX_BF16_ACT))
        return STYLEX_EINVAL;
    ConvKParams p;
    bwd_data_params(p, sh);
    p.a = (const float*)dy;
    p.w = w_bwd;
    p.y = (float*)dx;
    p.act_bf16 = precision == STYLEX_BF16_ACT;
    if (p.act_bf16) precision = STYLEX_BF16;
    if (precision == STYLEX_BF16 && p.KH == 3 && p.stride == 1 && p.pad == 1) {
        // dx = conv3x3(dy, flipped taps): forward-gather form, eligible for the LDS-halo kernel
        ConvKParams q = p;
        q.flip_taps = 1;
        q.flags = flags;
        if (epi) {
            q.a_scale = epi->in_scale;
            q.out_scale = epi->out_scale;
            q.residual = (const float*)epi->residual;  // gate tensor (STYLEX_EPI_GATE)
            q.res_scale = epi->res_scale;
            q.s2d_c = (int)epi->s2d_c;
            if (flags & STYLEX_EPI_GATE_MASK) q.gate_mask = (const unsigned char*)epi->mask;
            if (q.s2d_c && (q.N != 4 * q.s2d_c || q.s2d_c % 64)) return STYLEX_EINVAL;
        }
        if (!((flags & STYLEX_EPI_OSCALE) && !q.out_scale)) {
            double fl = 2.0 * (double)sh[0] * sh[9] * sh[10] * (double)sh[4] * sh[3] * sh[5] * sh[6] / (q.s2d_c ? 4.0 : 1.0);
            ScopedTimer tmh(1, fl, conv_bytes(sh, q.act_bf16 ? STYLEX_BF16_ACT : precision, false), (hipStream_t)stream, sh, q.s2d_c);
            int rc = stylex_launch_halo(q, (hipStream_t)stream);
            if (rc != STYLEX_NOT_APPLICABLE) return rc;
            if (tmh.on) {  // not applicable: nothing was launched; fall through to the generic kernel
                (void)hipEventDestroy(tmh.t.start);
                (void)hipEventDestroy(tmh.t.stop);
                tmh.on = false;
            }
        }
    }
    if (flags & STYLEX_EPI_GATE_MASK) return STYLEX_EINVAL;  // only the LDS-DMA kernel reads masks (stylex_conv_mask_supported)
    p.transposed = 1;
    p.phase_major = (p.stride == 2 && (p.Ho % 2 == 0) && (p.Wo % 2 == 0)) ? 1 : 0;
    p.flags = flags;
    if (epi) {
        p.a_scale = epi->in_scale;    // [B][N_f] applied to dy (demodulation coefficient)
        p.out_scale = epi->out_scale; // [B][C_f] applied to dx (modulation)
        p.residual = (const float*)epi->residual;  // gate tensor (STYLEX_EPI_GATE)
        p.res_scale = epi->res_scale;
    }
    if ((flags & STYLEX_EPI_OSCALE) && !p.out_scale) return STYLEX_EINVAL;
    // algorithmic FLOPs of a data gradient = those of the forward conv
    double flops = 2.0 * (double)sh[0] * sh[9] * sh[10] * (double)sh[4] * sh[3] * sh[5] * sh[6];
    ScopedTimer tm(1, flops, conv_bytes(sh, p.act_bf16 ? STYLEX_BF16_ACT : precision, false), (hipStream_t)stream, sh, 0);
    return stylex_launch_igemm(p, precision, workspace, workspace_bytes, (hipStream_t)stream);
}

static void wgrad_params(ConvKParams& p, const int64_t* sh) {
    fill_common(p, sh);
    p.Hi = (int)sh[1];
    p.Wi = (int)sh[2];
    p.Ck = (int)sh[3];
    p.N = (int)sh[4];
    p.Ho = (int)sh[9];
    p.Wo = (int)sh[10];
    p.M = p.B * p.Ho * p.Wo;
}

int64_t stylex_conv2d_bwd_weight_workspace_bytes(const int64_t* sh) {
    if (!conv_shape_ok(sh)) return STYLEX_EINVAL;
    ConvKParams p;
    wgrad_params(p, sh);
    int tn, tc, splits;
    long len;
    stylex_wgrad_plan(p, &tn, &tc, &splits, &len);
    {  // pipelined LDS-DMA plan (bf16 activations; with / without a per-sample x scale)
        static const float dummy_scale[4] = {0.f, 0.f, 0.f, 0.f};
        for (int sc = 0; sc < 2; ++sc) {
            ConvKParams q = p;
            q.act_bf16 = 1;
            q.a = q.a2 = nullptr;
            q.a_scale = sc ? dummy_scale : nullptr;
            if (stylex_wgrad_pipe_applicable(q)) {
                int sl, tps, blocks;
                stylex_wgrad_pipe_plan(q, &sl, &tps, &blocks);
                if (sl > splits) splits = sl;
            }
        }
    }
    if (p.Ck % 8 == 0 && p.N % 8 == 0) {  // tr kernel plans: the register-staged one and (bf16 activations) the LDS-DMA one
        for (int abf = 0; abf < 2; ++abf) {
            ConvKParams q = p;
            q.act_bf16 = abf;
            int mode, ts;
            long tl;
            stylex_wgrad_tr_plan(q, &mode, &ts, &tl);
            if (ts > splits) splits = ts;
        }
    }
    // (the space-to-depth form of the pipelined kernel has the plan of its 4C-channel stride-1 shape, which the loop above
    // covers: s2d_c narrows the applicability test only)
    // + room for the per-split bias sums of stylex_conv2d_bwd_weight_bias
    return (int64_t)splits * p.N * p.Ck * p.KH * p.KW * (int64_t)sizeof(float) + (int64_t)splits * p.N * (int64_t)sizeof(float);
}

int stylex_conv2d_bwd_weight(const void* x, const void* dy, float* dw, void* workspace, int64_t workspace_bytes,
                             const int64_t* sh, const float* x_scale, const float* dy_scale, int s2d_c, int precision,
                             void* stream) {
    return stylex_conv2d_bwd_weight_bias(x, dy, dw, nullptr, nullptr, workspace, workspace_bytes, sh, x_scale, dy_scale, s2d_c,
                                         precision, stream);
}

int stylex_conv2d_bwd_weight_bias(const void* x, const void* dy, float* dw, float* db, int* db_written, void* workspace,
                                  int64_t workspace_bytes, const int64_t* sh, const float* x_scale, const float* dy_scale,
                                  int s2d_c, int precision, void* stream) {
    return stylex_conv2d_bwd_weight_ex(x, dy, dw, db, db_written, workspace, workspace_bytes, sh, x_scale, dy_scale, s2d_c, 1.f, 0,
                                       precision, stream);
}

int stylex_conv2d_bwd_weight_ex(const void* x, const void* dy, float* dw, float* db, int* db_written, void* workspace,
                                int64_t workspace_bytes, const int64_t* sh, const float* x_scale, const float* dy_scale,
                                int s2d_c, float out_scale, int accumulate, int precision, void* stream) {
    if (db_written) *db_written = 0;
    if (db && !db_written) return STYLEX_EINVAL;
    if (!x || !dy || !dw || !workspace || !conv_shape_ok(sh) || sh[7] > 2) return STYLEX_EINVAL;
    if (precision != STYLEX_F32 && precision != STYLEX_BF16 && precision != STYLEX_BF16_ACT) return STYLEX_EINVAL;
    if (workspace_bytes < stylex_conv2d_bwd_weight_workspace_bytes(sh)) return STYLEX_EWORKSPACE;
    ConvKParams p;
    wgrad_params(p, sh);
    p.a = (const float*)x;
    p.a_scale = x_scale;
    p.a2 = (const float*)dy;
    p.a2_scale = dy_scale;
    p.act_bf16 = precision == STYLEX_BF16_ACT;
    if (p.act_bf16) precision = STYLEX_BF16;
    p.s2d_c = s2d_c;
    p.wg_scale = out_scale;
    p.wg_accumulate = accumulate ? 1 : 0;
    if (s2d_c && (p.Ck != 4 * s2d_c || s2d_c % 64 || p.KH != 3 || p.stride != 1)) return STYLEX_EINVAL;
    double flops = 2.0 * p.M * (double)p.N * p.Ck * p.KH * p.KW / (s2d_c ? 4.0 : 1.0);
    ScopedTimer tm(2, flops, conv_bytes(sh, p.act_bf16 ? STYLEX_BF16_ACT : precision, true), (hipStream_t)stream, sh, s2d_c);
    return stylex_launch_wgrad(p, (float*)workspace, dw, precision, (hipStream_t)stream, db, db_written);
}

int stylex_conv2d_bwd_weight_s2d_supported(const int64_t* sh, int s2d_c, int precision) {
    if (!conv_shape_ok(sh) || precision != STYLEX_BF16_ACT || s2d_c <= 0) return 0;
    ConvKParams p;
    wgrad_params(p, sh);
    p.act_bf16 = 1;
    p.s2d_c = s2d_c;
    return stylex_wgrad_pipe_applicable(p) ? 1 : 0;
}

int stylex_conv2d_bwd_weight_s2d(const void* x2, const void* dy, float* dw, void* workspace, int64_t workspace_bytes,
                                 const int64_t* sh, int s2d_c, float out_scale, int accumulate, int precision, void* stream) {
    if (!x2 || !dy || !dw || !workspace || !conv_shape_ok(sh) || precision != STYLEX_BF16_ACT || s2d_c <= 0) return STYLEX_EINVAL;
    if (workspace_bytes < stylex_conv2d_bwd_weight_workspace_bytes(sh)) return STYLEX_EWORKSPACE;
    ConvKParams p;
    wgrad_params(p, sh);
    p.a = (const float*)x2;
    p.a2 = (const float*)dy;
    p.act_bf16 = 1;
    p.s2d_c = s2d_c;
    p.wg_scale = out_scale;
    p.wg_accumulate = accumulate ? 1 : 0;
    if (!stylex_wgrad_pipe_applicable(p)) return STYLEX_EINVAL;
    double flops = 2.0 * p.M * (double)p.N * p.Ck * 9 / 4.0;
    ScopedTimer tm(2, flops, conv_bytes(sh, STYLEX_BF16_ACT, true), (hipStream_t)stream, sh, s2d_c);
    return stylex_launch_wgrad_s2d_folded(p, (float*)workspace, dw, (hipStream_t)stream);
}

int stylex_pack_weight_s2d(const float* w, void* wf, void* wb, const int64_t* sh, void* stream) {
    if (!w || sh[0] < 1 || sh[1] < 1 || sh[2] != 3 || sh[3] != 3) return STYLEX_EINVAL;
    return stylex_launch_pack_s2d(w, wf, wb, (int)sh[0], (int)sh[1], (hipStream_t)stream);
}

int stylex_fold_weight_grad_s2d(const float* dw2, float* dw, const int64_t* sh, void* stream) {
    if (!dw2 || !dw || sh[0] < 1 || sh[1] < 1 || sh[2] != 3 || sh[3] != 3) return STYLEX_EINVAL;
    return stylex_launch_fold_s2d(dw2, dw, (int)sh[0], (int)sh[1], (hipStream_t)stream);
}

}  // extern "C"
