// conv_wgrad_tr.hip — general weight gradient (any kernel size / stride / padding) for bf16 activations on
// gfx950: the small-spatial 3x3 layers (<= 8x8, where the resident-halo kernel has no tile to work with), the
// 1x1 / stride-2 residual convolutions, the non-space-to-depth stride-2 convs and the 3-channel first layers.
//
//   dW[n][tap][c] = sum_m dy[m][n] * x[pix(m) + tap][c]          m = (b, oh, ow)
//
// GEMM view per tap: D[n][c] += A[n][k=m] * B[k=m][c]; both operands are pixel-major in memory (NHWC), i.e.
// k is the slow axis.  As in conv_wgrad_halo.hip the transpose happens in the LDS read: the staged tiles stay
// pixel-major (raw 16-byte slots, no conversion, conflict-free 16-byte stores) in 32-channel panels of 64-byte
// pixel rows, and ds_read_b64_tr_b16 hands every lane 4 consecutive pixels of its channel.  The previous
// generic kernel converted to fp32 and transposed with 2-byte LDS stores.
//
// A block owns a (2*TN*32) x (2*TC*32) tile of one tap (4 waves as 2x2), walks a contiguous pixel range
// (split-K over pixels, partials reduced in fixed order by wgrad_reduce_kernel -> deterministic), K-tiles of 64
// pixels, double-buffered in LDS with the global loads of tile k+1 in flight under the MFMAs of tile k.
//
// C8 mode (C_in == 8: the RGB input padded to one 16-byte slot): the 8 channels of all T taps are flattened
// into one "channel" axis of T*8 (slot q of a pixel row = tap q), so one block produces all taps instead of T
// blocks each using 8 of 64 MFMA columns.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<unsigned*>(&r);
}

typedef __attribute__((address_space(3))) s16x4* lds_s4_ptr;

__device__ __forceinline__ bf16x8 tr_read8(const char* base, int byte_off) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + byte_off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(base + byte_off + 4 * 64));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

constexpr int BKP = 64;                  // pixels per K-tile
constexpr int PANEL = BKP * 64 + 64;     // one 32-channel panel of a K-tile (+64 B skew between panels)

template <int TN_, int TC_>
struct TrGeom {
    static constexpr int BNn = 2 * TN_ * 32, BC = 2 * TC_ * 32;
    static constexpr int NPA = BNn / 32, NPB = BC / 32;
    static constexpr int STAGE = (NPA + NPB) * PANEL;
    static constexpr int SMEM_BYTES = 2 * STAGE;
};

template <int TN_, int TC_, bool C8>
__global__ __launch_bounds__(256) void conv_wgrad_tr_kernel(ConvKParams p) {
    typedef TrGeom<TN_, TC_> G;
    constexpr int BNn = G::BNn, BC = G::BC, NPA = G::NPA, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int T = p.KH * p.KW, Teff = C8 ? 1 : T;
    const int N = p.N, C = p.Ck;
    const int n_tiles = (N + BNn - 1) / BNn, c_tiles = C8 ? 1 : (C + BC - 1) / BC;
    const int per_split = n_tiles * c_tiles * Teff;

    const int bid = blockIdx.x;
    const int tile = bid % per_split, split = bid / per_split;
    const int tap = tile % Teff;
    const int n0 = ((tile / Teff) / c_tiles) * BNn;
    const int c0 = ((tile / Teff) % c_tiles) * BC;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    // pixel indices fit 32 bits (M = B*Ho*Wo); 64-bit divisions in the staging loop cost more than the loads
    const int m_begin = (int)((long)split * p.split_len);
    const int m_end = (int)min((long)p.M, (long)m_begin + p.split_len);
    const int nk = (m_end - m_begin + BKP - 1) / BKP;
    const int hw = p.Ho * p.Wo;
    // pixel -> (b, oh, ow): two unsigned divisions per staged 16-byte slot were most of this kernel's instructions (PMC, round 5:
    // 25 VALU instructions per MFMA at 8x8 px); every StylEx grid is a power of two, so they are shifts behind a uniform branch
    const bool pow2 = (hw & (hw - 1)) == 0 && (p.Wo & (p.Wo - 1)) == 0;
    const int sh_hw = __builtin_ctz((unsigned)hw), sh_w = __builtin_ctz((unsigned)p.Wo);

    constexpr int A_SLOTS = BNn / 8, B_SLOTS = BC / 8;
    constexpr int A_PER = BKP * A_SLOTS / 256, B_PER = BKP * B_SLOTS / 256;
    uint4 ra[A_PER], rb[B_PER];
    const unsigned short* dy = reinterpret_cast<const unsigned short*>(p.a2);
    const unsigned short* xs = reinterpret_cast<const unsigned short*>(p.a);

    auto scaled = [&](uint4 v, const float* sc) -> uint4 {
        float4 a0 = *reinterpret_cast<const float4*>(sc), a1 = *reinterpret_cast<const float4*>(sc + 4);
        float4 f0 = act_unpack4(make_uint2(v.x, v.y)), f1 = act_unpack4(make_uint2(v.z, v.w));
        v.x = pack_bf16(f0.x * a0.x, f0.y * a0.y); v.y = pack_bf16(f0.z * a0.z, f0.w * a0.w);
        v.z = pack_bf16(f1.x * a1.x, f1.y * a1.y); v.w = pack_bf16(f1.z * a1.z, f1.w * a1.w);
        return v;
    };

    auto load_tile = [&](int kt) {
        const int mb = m_begin + kt * BKP;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int idx = tid + 256 * j;
            const int pr = idx / A_SLOTS, q = idx % A_SLOTS;
            const int m = mb + pr;
            const int n = n0 + q * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (m < m_end && n < N) {
                v = *reinterpret_cast<const uint4*>(dy + (long)m * N + n);
                if (p.a2_scale) v = scaled(v, p.a2_scale + (long)(pow2 ? (unsigned)m >> sh_hw : (unsigned)m / (unsigned)hw) * N + n);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int idx = tid + 256 * j;
            const int pr = idx / B_SLOTS, q = idx % B_SLOTS;
            const int m = mb + pr;
            int c, th, tw;
            bool ok = m < m_end;
            if (C8) {
                c = 0;
                th = q / p.KW;
                tw = q - th * p.KW;
                ok = ok && q < T;
            } else {
                c = c0 + q * 8;
                th = kh;
                tw = kw;
                ok = ok && c < C;
            }
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ok) {
                int b, oh;
                if (pow2) {
                    b = (int)((unsigned)m >> sh_hw);
                    oh = (int)(((unsigned)m & (unsigned)(hw - 1)) >> sh_w);
                } else {
                    b = (int)((unsigned)m / (unsigned)hw);
                    oh = (int)((unsigned)(m - b * hw) / (unsigned)p.Wo);
                }
                const int r = m - b * hw;
                const int ow = r - oh * p.Wo;
                const int ih = oh * p.stride + th - p.pad, iw = ow * p.stride + tw - p.pad;
                if (ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi) {
                    v = *reinterpret_cast<const uint4*>(xs + (((long)b * p.Hi + ih) * p.Wi + iw) * C + c);
                    if (p.a_scale) v = scaled(v, p.a_scale + (long)b * C + c);
                }
            }
            rb[j] = v;
        }
    };

    auto store_tile = [&](int st) {
        char* base = smem + st * STAGE;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int idx = tid + 256 * j;
            const int pr = idx / A_SLOTS, q = idx % A_SLOTS;
            *reinterpret_cast<uint4*>(base + (q >> 2) * PANEL + pr * 64 + (q & 3) * 16) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int idx = tid + 256 * j;
            const int pr = idx / B_SLOTS, q = idx % B_SLOTS;
            *reinterpret_cast<uint4*>(base + (NPA + (q >> 2)) * PANEL + pr * 64 + (q & 3) * 16) = rb[j];
        }
    };

    f32x16 acc[TN_][TC_];
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TC_; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // tr-read lane addressing (conv_wgrad_halo.hip): group g = lane>>4 -> channel block (g&1)*16, k half (g>>1)*8
    const int i16 = lane & 15, g = lane >> 4;
    const int lane_off = ((g >> 1) * 8 + (i16 >> 2)) * 64 + ((g & 1) * 16 + (i16 & 3) * 4) * 2;

    auto compute = [&](int st) {
        const char* a_base = smem + st * STAGE + (wn * TN_) * PANEL + lane_off;
        const char* b_base = smem + st * STAGE + (NPA + wc * TC_) * PANEL + lane_off;
#pragma unroll
        for (int ks = 0; ks < BKP / 16; ++ks) {
            bf16x8 av[TN_], bv[TC_];
#pragma unroll
            for (int i = 0; i < TN_; ++i) av[i] = tr_read8(a_base + i * PANEL, ks * 16 * 64);
#pragma unroll
            for (int j = 0; j < TC_; ++j) bv[j] = tr_read8(b_base + j * PANEL, ks * 16 * 64);
#pragma unroll
            for (int i = 0; i < TN_; ++i)
#pragma unroll
                for (int j = 0; j < TC_; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0, st = 0; kt < nk; ++kt, st ^= 1) {
        if (kt + 1 < nk) load_tile(kt + 1);
        compute(st);
        if (kt + 1 < nk) store_tile(st ^ 1);
        __syncthreads();
    }

    // partial[split][n][tap][c]; D[i=n][j=c]: col j = lane&31, row i = (r&3)+8*(r>>2)+4*(lane>>5)
    const int lj = lane & 31, lh = lane >> 5;
    float* out = p.y + (long)split * N * T * C;
#pragma unroll
    for (int j = 0; j < TC_; ++j) {
        const int cj = c0 + (wc * TC_ + j) * 32 + lj;
        if (C8 ? (cj >= T * 8) : (cj >= C)) continue;
#pragma unroll
        for (int i = 0; i < TN_; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * TN_ + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < N) out[C8 ? ((long)n * T * 8 + cj) : (((long)n * T + tap) * C + cj)] = acc[i][j][r];
            }
    }
}

template <int TN_, int TC_, bool C8>
int launch_tr(const ConvKParams& p, int blocks, hipStream_t s) {
    auto k = conv_wgrad_tr_kernel<TN_, TC_, C8>;
    constexpr int smem_bytes = TrGeom<TN_, TC_>::SMEM_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           smem_bytes);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    stylex_note_kernel("conv_wgrad_tr_kernel<%d, %d, %s>", TN_, TC_, C8 ? "true" : "false");
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), smem_bytes, s, p);
    return (int)hipGetLastError();
}

}  // namespace

bool stylex_wgrad_tr_applicable(const ConvKParams& p) {
    if (!p.act_bf16 || p.Ck % 8 != 0 || p.N % 8 != 0 || p.s2d_c) return false;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.a2) & 15)) return false;
    if (p.a_scale && (reinterpret_cast<uintptr_t>(p.a_scale) & 15)) return false;
    if (p.a2_scale && (reinterpret_cast<uintptr_t>(p.a2_scale) & 15)) return false;
    if (p.Ck == 8 && p.KH * p.KW > 16) return false;
    return true;
}

// tile shape + split-K plan; mode: 0 = 128x128, 1 = 64x64, 2 = C8 (64 x 128 flattened taps)
void stylex_wgrad_tr_plan(const ConvKParams& p, int* mode, int* splits, long* split_len) {
    const int T = p.KH * p.KW;
    long tiles;
    if (p.Ck == 8) {
        *mode = 2;
        tiles = (p.N + 63) / 64;
    } else if (p.N > 64 && p.Ck > 64) {
        *mode = 0;
        tiles = (long)((p.N + 127) / 128) * ((p.Ck + 127) / 128) * T;
    } else {
        *mode = 1;
        tiles = (long)((p.N + 63) / 64) * ((p.Ck + 63) / 64) * T;
    }
    long want = (768 + tiles - 1) / tiles;                      // ~3 blocks per CU
    long max_by_len = ((long)p.M + 2 * BKP - 1) / (2 * BKP);    // at least 2 K-tiles per split
    long sp = want < 1 ? 1 : want;
    if (sp > max_by_len) sp = max_by_len;
    if (sp < 1) sp = 1;
    if (sp > 1024) sp = 1024;
    long len = (((long)p.M + sp - 1) / sp + BKP - 1) / BKP * BKP;
    sp = ((long)p.M + len - 1) / len;
    *splits = (int)sp;
    *split_len = len;
}

int stylex_launch_wgrad_tr(ConvKParams p, float* partial, hipStream_t s, int* splits_out) {
    int mode, splits;
    long len;
    stylex_wgrad_tr_plan(p, &mode, &splits, &len);
    p.split_len = len;
    p.y = partial;
    *splits_out = splits;
    const int T = p.KH * p.KW;
    if (mode == 2) return launch_tr<1, 2, true>(p, ((p.N + 63) / 64) * splits, s);
    if (mode == 0) return launch_tr<2, 2, false>(p, ((p.N + 127) / 128) * ((p.Ck + 127) / 128) * T * splits, s);
    return launch_tr<1, 1, false>(p, ((p.N + 63) / 64) * ((p.Ck + 63) / 64) * T * splits, s);
}
