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
#include <stdlib.h>

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

// ---- round 5: the 128 x 128 tile of the kernel above with LDS-DMA staging (conv_gather_line_kernel's lesson: at <= 8 px the
// register-staged version spends 25 VALU instructions per MFMA on addresses, bounds and ds_writes — PMC: matrix pipe 10 % busy).
// Same tile, taps-per-block, partial layout and LDS image (pixel-major 32-channel panels of 64-byte rows, read with
// ds_read_b64_tr_b16); 8 waves; a K stage = 64 pixels = 32 DMA instructions of 16 rows x 64 B (waves 0-3: the dy panels,
// 4-7: the x panels), four-stage ring with three stages in flight, counted vmcnt + raw barrier; everything a DMA needs per
// lane is a 32-bit offset in a register (dy: constant, the stage term is the scalar offset; x: pixel decode by shifts,
// out-of-range offset for padding).  Conditions: bf16, stride 1 (or the stride-2 3x3 down conv), power-of-two grid, N and C
// multiples of 128, no scales.
constexpr int DSTAGES = 4, DPIECES = 4;
constexpr int DSTAGE_BYTES = 8 * PANEL;

struct TrOps {  // the six transpose reads of one 16-pixel k-step: dy fragment (2 x b64), two x fragments
    s16x4 a[2];
    s16x4 b[2][2];
};
template <int OFF>
__device__ __forceinline__ void tr_asm(s16x4& dst, int addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int KS>
__device__ __forceinline__ void tr_issue(TrOps& o, int a_addr, int b_addr) {
    tr_asm<KS * 16 * 64>(o.a[0], a_addr);
    tr_asm<KS * 16 * 64 + 4 * 64>(o.a[1], a_addr);
    tr_asm<KS * 16 * 64>(o.b[0][0], b_addr);
    tr_asm<KS * 16 * 64 + 4 * 64>(o.b[0][1], b_addr);
    tr_asm<PANEL + KS * 16 * 64>(o.b[1][0], b_addr);
    tr_asm<PANEL + KS * 16 * 64 + 4 * 64>(o.b[1][1], b_addr);
}
template <int N>
__device__ __forceinline__ void tr_wait(TrOps& o) {  // all but the N newest LDS reads have landed
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(o.a[0]), "+v"(o.a[1]), "+v"(o.b[0][0]), "+v"(o.b[0][1]), "+v"(o.b[1][0]), "+v"(o.b[1][1])
                 : "n"(N));
}
__device__ __forceinline__ bf16x8 cat8(s16x4 lo, s16x4 hi) {
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__global__ __launch_bounds__(512, 1) void conv_wgrad_tr_dma_kernel(ConvKParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.KH * p.KW;
    const int N = __builtin_amdgcn_readfirstlane(p.N), C = __builtin_amdgcn_readfirstlane(p.Ck);
    const int n_tiles = N >> 7, c_tiles = C >> 7;
    const int per_split = n_tiles * c_tiles * T;
    const int bid = blockIdx.x;
    const int tile = bid % per_split, split = bid / per_split;
    const int tap = tile % T;
    const int n0 = ((tile / T) / c_tiles) << 7;
    const int c0 = ((tile / T) % c_tiles) << 7;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int m_begin = (int)((long)split * p.split_len);
    const int m_end = (int)min((long)p.M, (long)m_begin + p.split_len);
    const int nk = (m_end - m_begin + BKP - 1) / BKP;
    const int Wo = __builtin_amdgcn_readfirstlane(p.Wo), hw = __builtin_amdgcn_readfirstlane(p.Ho * p.Wo);
    const int Hi = __builtin_amdgcn_readfirstlane(p.Hi), Wi = __builtin_amdgcn_readfirstlane(p.Wi);
    const int sh_hw = __builtin_ctz((unsigned)hw), sh_w = __builtin_ctz((unsigned)Wo);
    const int dh = kh - p.pad, dw = kw - p.pad, strd = __builtin_amdgcn_readfirstlane(p.stride);

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a2), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, 0x7ffffff0, 0x00020000);
    constexpr unsigned OOBV = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_void_ptr;

    // ---- staging: DMA instruction `it` of a stage = pixel rows it * 16 + (lane >> 2) of this wave's panel, slot lane & 3
    const bool stage_x = wave >= 4;
    const int panel = wave & 3;
    const int chan = panel * 32 + (lane & 3) * 8;
    int mrow[DPIECES];
    unsigned a_off[DPIECES];
#pragma unroll
    for (int it = 0; it < DPIECES; ++it) {
        mrow[it] = m_begin + it * 16 + (lane >> 2);
        a_off[it] = ((unsigned)mrow[it] * (unsigned)N + (unsigned)(n0 + chan)) * 2u;
    }
    int i_kt = 0;  // the K-tile the DMA stream fetches next
    auto issue_stage = [&](int rs) {
        const int dst = rs * DSTAGE_BYTES + ((stage_x ? 4 : 0) + panel) * PANEL;
        const int mshift = i_kt * BKP;
        if (!stage_x) {
            const unsigned soff = (unsigned)mshift * (unsigned)N * 2u;
#pragma unroll
            for (int it = 0; it < DPIECES; ++it) {
                const unsigned v = mrow[it] + mshift < m_end ? a_off[it] : OOBV;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_ptr)(smem + dst + it * 1024), 16, v, soff, 0, 0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < DPIECES; ++it) {
                const int m = mrow[it] + mshift;
                const int b = (int)((unsigned)m >> sh_hw), r = m & (hw - 1);
                const int ih = (r >> sh_w) * strd + dh, iw = (r & (Wo - 1)) * strd + dw;
                const bool ok = m < m_end && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
                const unsigned v = ok ? ((unsigned)((b * Hi + ih) * Wi + iw) * (unsigned)C + (unsigned)(c0 + chan)) * 2u : OOBV;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_ptr)(smem + dst + it * 1024), 16, v, 0u, 0, 0);
            }
        }
        ++i_kt;
    };

    // ---- MFMA roles: wave (wn, wc) = (wave >> 1, wave & 1): dy panel wn (32 n) x x panels 2 wc, 2 wc + 1 (64 c)
    const int wn = wave >> 1, wc = wave & 1;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int i16 = lane & 15, g = lane >> 4;
    const int lane_off = ((g >> 1) * 8 + (i16 >> 2)) * 64 + ((g & 1) * 16 + (i16 & 3) * 4) * 2;
    const int lds0 = (int)(unsigned)(uintptr_t)(lds_void_ptr)smem;  // LDS byte address of the dynamic allocation

    for (int s = 0; s < 3 && s < nk; ++s) issue_stage(s);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPIECES) : "memory");
        else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (raw: __syncthreads() would drain the ring)
        asm volatile("" ::: "memory");
        if (kt + 3 < nk) issue_stage((kt + 3) % DSTAGES);
        // operand reads as asm (hipcc puts an s_waitcnt vmcnt(0) in front of every LDS read it can see behind a load-to-LDS,
        // which would drain the ring), one k-step ahead of the MFMAs
        const int a_addr = lds0 + (kt % DSTAGES) * DSTAGE_BYTES + wn * PANEL + lane_off;
        const int b_addr = lds0 + (kt % DSTAGES) * DSTAGE_BYTES + (4 + wc * 2) * PANEL + lane_off;
        TrOps o[2];
        tr_issue<0>(o[0], a_addr, b_addr);
#pragma unroll
        for (int ks = 0; ks < BKP / 16; ++ks) {
            if (ks + 1 < BKP / 16) {
                if (ks == 0) tr_issue<1>(o[1], a_addr, b_addr);
                if (ks == 1) tr_issue<2>(o[0], a_addr, b_addr);
                if (ks == 2) tr_issue<3>(o[1], a_addr, b_addr);
                tr_wait<6>(o[ks & 1]);
            } else {
                tr_wait<0>(o[ks & 1]);
            }
            const bf16x8 av = cat8(o[ks & 1].a[0], o[ks & 1].a[1]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, cat8(o[ks & 1].b[0][0], o[ks & 1].b[0][1]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, cat8(o[ks & 1].b[1][0], o[ks & 1].b[1][1]), acc[1], 0, 0, 0);
        }
    }

    // partial[split][n][tap][c]; D[i=n][j=c]: col j = lane&31, row i = (r&3)+8*(r>>2)+4*(lane>>5)
    const int lj = lane & 31, lh = lane >> 5;
    float* out = p.y + (long)split * N * T * C;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cj = c0 + (wc * 2 + j) * 32 + lj;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((long)n * T + tap) * C + cj] = acc[j][r];
        }
    }
}

static bool tr_dma_applicable(const ConvKParams& p) {
    const char* env = getenv("STYLEX_WGRAD_TR_DMA");  // read per launch: A/B tests toggle it in-process
    if (env && env[0] == '0') return false;
    const int hw = p.Ho * p.Wo;
    if (!p.act_bf16 || p.a_scale || p.a2_scale || p.s2d_c) return false;
    if (!((p.KH == 3 && p.KW == 3 && p.pad == 1) || (p.KH == 1 && p.KW == 1 && p.pad == 0))) return false;
    // stride 1, or the 3x3 / stride-2 down conv of the small blocks (16^2 -> 8^2 and below: x is [B][2 Ho][2 Wo][C])
    if (!((p.stride == 1 && p.Hi == p.Ho && p.Wi == p.Wo) || (p.stride == 2 && p.KH == 3 && p.Hi == 2 * p.Ho && p.Wi == 2 * p.Wo))) return false;
    if (p.N % 128 != 0 || p.Ck % 128 != 0 || (hw & (hw - 1)) || (p.Wo & (p.Wo - 1))) return false;
    if (hw > 64 && p.KH != 1) return false;  // larger 3x3 grids belong to the pipelined kernel; 1x1 convs (the residual convs) run here at any size
    if ((long)p.M * p.N * 2 >= (1l << 31) - 16 || (long)p.M * p.Ck * 2 >= (1l << 31) - 16) return false;
    return true;
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
    const bool dma = *mode == 0 && tr_dma_applicable(p);         // one 8-wave block per CU: ~1.5 blocks per CU, >= 8 K-tiles each
    long want = ((dma ? 384 : 768) + tiles - 1) / tiles;        // (else ~3 blocks per CU)
    long max_by_len = ((long)p.M + (dma ? 8 : 2) * BKP - 1) / ((dma ? 8 : 2) * BKP);  // at least 2 K-tiles per split
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
    if (mode == 0 && tr_dma_applicable(p)) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_tr_dma_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, DSTAGES * DSTAGE_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_done = true;
        }
        stylex_note_kernel("conv_wgrad_tr_dma_kernel");
        hipLaunchKernelGGL(conv_wgrad_tr_dma_kernel, dim3((p.N / 128) * (p.Ck / 128) * T * splits), dim3(512), DSTAGES * DSTAGE_BYTES, s, p);
        return (int)hipGetLastError();
    }
    if (mode == 0) return launch_tr<2, 2, false>(p, ((p.N + 127) / 128) * ((p.Ck + 127) / 128) * T * splits, s);
    return launch_tr<1, 1, false>(p, ((p.N + 63) / 64) * ((p.Ck + 63) / 64) * T * splits, s);
}
