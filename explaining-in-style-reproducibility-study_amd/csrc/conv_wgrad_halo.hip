// conv_wgrad_halo.hip — weight gradient of a 3x3 / stride-1 / pad-1 convolution on gfx950 (bf16 MFMA,
// fp32 accumulate), with the activation tile + halo and the dy tile resident in LDS.
//
//   dW[n][tap][c] = sum_{b,y,x} dy[b,y,x,n] * xin[b,y+kh-1,x+kw-1,c]
//
// GEMM view: D[i=n][j=c] += A[n][k=pixel] * B[k=pixel][c].  Both operands are stored pixel-major in
// memory (NHWC), i.e. k is the SLOW axis, while an MFMA lane needs 8 consecutive k of one row.  The
// transpose is done by the LDS transpose-read ds_read_b64_tr_b16: within a 16-lane group, lane i
// supplies the address of 4 contiguous bf16; lane l receives element (l&3) of rows 4j+(l>>2), j=0..3
// (measured with tools/tr_probe.hip).  Giving lane i the address [pixel (i>>2)][channel 4*(i&3)]
// returns to lane l the 4 consecutive pixels of channel l — and a 3x3 tap shift is just a different
// starting ROW, so all 9 taps read the same resident halo with no realignment.
//
// A block owns a 64(n) x 64(c) x 9-tap accumulator (144 fp32 registers per lane, 4 waves as 2x2 of
// 32x32) and walks a contiguous range of 8x32-pixel tiles (split-K over tiles); per tile it stages
// dy [256 px][64 n] and the x halo [340 px][64 c] as bf16 (fp32 -> bf16 on the way, per-sample
// modulation/demodulation scales folded in), in 32-channel panels of unpadded 64-byte pixel rows:
// a 32-lane tr-read then covers 4 rows x 64 B = all 64 banks exactly once (conflict free), and the
// 8-byte staging stores are conflict free as well.  Partials go to workspace[split][n][tap][c] and
// are reduced in fixed order by wgrad_reduce_kernel (deterministic).
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

__device__ __forceinline__ bf16x8 tr_read8(const char* smem_base, int byte_off, int row_pitch_bytes) {
    // two transpose reads: pixels k..k+3 and k+4..k+7 of this lane's channel
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(smem_base + byte_off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(smem_base + byte_off + 4 * row_pitch_bytes));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

constexpr int PIX_ROW = 64;  // bytes per pixel row of a 32-channel panel
// pixel tile: 8x32 (10x34 halo) or, for 16-pixel-wide images, 16x16 (18x18 halo)
template <int TW>
struct Geom {
    static constexpr int TH = 256 / TW, HWD = TW + 2, NP = (TH + 2) * HWD;
    // panel strides carry a 64-byte skew so that the two panels a 16-lane staging store touches fall on
    // different bank halves
    static constexpr int DY_PANEL = 256 * PIX_ROW + 64;  // 16 KiB (+skew)
    static constexpr int X_PANEL = NP * PIX_ROW + 64;    // ~21 KiB (+skew)
    static constexpr int X_OFF = 2 * DY_PANEL;
    static constexpr int SMEM_BYTES = 2 * DY_PANEL + 2 * X_PANEL;  // ~75 KiB -> 2 blocks / CU
};

// BIAS (MASK must not contain tap 0): acc[0] += dy x ONES, i.e. every column of acc[0] = sum over the tile's pixels of
// dy[., n] — the bias gradient, for the price of one MFMA per k-step and no extra pass over dy.
template <int TW, unsigned MASK, bool BIAS = false>
__device__ __forceinline__ void compute_tile(f32x16 (&acc)[9], const char* a_base, const char* b_base) {
    constexpr int HWD = Geom<TW>::HWD, KS_PER_ROW = TW / 16;
    static_assert(!BIAS || !(MASK & 1u), "the bias sum lives in the accumulator of tap 0");
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};  // bf16 1.0
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int r = ks / KS_PER_ROW, pw0 = (ks % KS_PER_ROW) * 16;
        bf16x8 av = tr_read8(a_base, (r * TW + pw0) * PIX_ROW, PIX_ROW);
        if (BIAS) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, ones, acc[0], 0, 0, 0);
        bf16x8 bv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if ((MASK >> t) & 1u) bv[t] = tr_read8(b_base, ((r + t / 3) * HWD + pw0 + t % 3) * PIX_ROW, PIX_ROW);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if ((MASK >> t) & 1u) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[t], acc[t], 0, 0, 0);
    }
}

template <int TW, bool ABF, bool S2D>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_halo_kernel(ConvKParams p) {
    constexpr int TH = Geom<TW>::TH, HWD = Geom<TW>::HWD, NP = Geom<TW>::NP;
    constexpr int DY_PANEL = Geom<TW>::DY_PANEL, X_PANEL = Geom<TW>::X_PANEL, X_OFF = Geom<TW>::X_OFF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int H = p.Ho, W = p.Wo, C = p.Ck, N = p.N;
    const int n_tiles = (N + 63) / 64, c_tiles = (C + 63) / 64;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int total_tiles = p.B * tiles_img;

    // XCD-aware order: the blocks of one split (they all stream the SAME pixel tiles, each taking its own
    // 64-channel slices) get consecutive logical ids on ONE XCD, so the re-reads hit that XCD's L2 instead of
    // being fetched by all eight (PMC: 473 MB fetched per launch before this remap).
    int bid = blockIdx.x;
    {
        int nblk = gridDim.x, q = nblk >> 3, rr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int ot = bid % (n_tiles * c_tiles);   // output tile (fastest: blocks sharing pixel tiles run together)
    const int split = bid / (n_tiles * c_tiles);
    const int n0 = (ot / c_tiles) * 64, c0 = (ot % c_tiles) * 64;
    const int t_begin = split * (int)p.split_len;
    const int t_end = min(total_tiles, t_begin + (int)p.split_len);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // space-to-depth form of a stride-2 conv: taps that are structurally zero for this c-tile's sub-position
    const unsigned tapmask = S2D ? stylex_s2d_tap_mask((c0 + wc * 32) / p.s2d_c) : 0x1ffu;

    // tr-read lane addressing (see header): group g = lane>>4 -> channel block (g&1)*16, k half (g>>1)*8
    const int i16 = lane & 15, g = lane >> 4;
    const int lane_off = ((g >> 1) * 8 + (i16 >> 2)) * PIX_ROW + ((g & 1) * 16 + (i16 & 3) * 4) * 2;

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / tiles_img;
        const int tt = tile - b * tiles_img;
        const int y0 = (tt / tiles_x) * TH, x0 = (tt % tiles_x) * TW;

        if (ABF) {
            // bf16 activations: raw 16-byte slots (8 channels), 8 slots per 64-channel pixel row, no conversion.
            // ALL global loads of the tile (8 dy + 11 halo per thread) are issued before the first LDS store so
            // that one memory latency covers the whole tile.
            const int q = tid & 7;
            const int nn = n0 + q * 8, cc = c0 + q * 8;
            constexpr int XS = (NP + 31) / 32;
            float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0, t0 = s0, t1 = s0;
            if (p.a2_scale && nn < N) {
                s0 = *reinterpret_cast<const float4*>(p.a2_scale + (long)b * N + nn);
                s1 = *reinterpret_cast<const float4*>(p.a2_scale + (long)b * N + nn + 4);
            }
            if (p.a_scale && cc < C) {
                t0 = *reinterpret_cast<const float4*>(p.a_scale + (long)b * C + cc);
                t1 = *reinterpret_cast<const float4*>(p.a_scale + (long)b * C + cc + 4);
            }
            auto scaled = [&](uint4 v, const float4& a0, const float4& a1) -> uint4 {
                float4 f0 = act_unpack4(make_uint2(v.x, v.y)), f1 = act_unpack4(make_uint2(v.z, v.w));
                v.x = pack_bf16(f0.x * a0.x, f0.y * a0.y); v.y = pack_bf16(f0.z * a0.z, f0.w * a0.w);
                v.z = pack_bf16(f1.x * a1.x, f1.y * a1.y); v.w = pack_bf16(f1.z * a1.z, f1.w * a1.w);
                return v;
            };
            auto load_x = [&](int it) -> uint4 {
                int hp = (tid >> 3) + 32 * it;
                int hh = hp / HWD, ww = hp - hh * HWD;
                int y = y0 - 1 + hh, x = x0 - 1 + ww;
                if (hp < NP && cc < C && y >= 0 && y < H && x >= 0 && x < W)
                    return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.a) + ((long)(b * H + y) * W + x) * C + cc);
                return make_uint4(0u, 0u, 0u, 0u);
            };
            auto store_x = [&](int it, uint4 v) {
                int hp = (tid >> 3) + 32 * it;
                if (hp < NP) {
                    if (p.a_scale) v = scaled(v, t0, t1);
                    *reinterpret_cast<uint4*>(smem + X_OFF + (q >> 2) * X_PANEL + hp * PIX_ROW + (q & 3) * 16) = v;
                }
            };
            constexpr int XH = (XS + 1) / 2;
            uint4 vd[8], vx[XH];
            // batch 1 in flight: 8 dy + first half of the halo
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int px = (tid >> 3) + 32 * it;
                int y = y0 + px / TW, x = x0 + (px % TW);
                vd[it] = make_uint4(0u, 0u, 0u, 0u);
                if (nn < N && y < H && x < W)
                    vd[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.a2) + ((long)(b * H + y) * W + x) * N + nn);
            }
#pragma unroll
            for (int it = 0; it < XH; ++it) vx[it] = load_x(it);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int px = (tid >> 3) + 32 * it;
                uint4 v = vd[it];
                if (p.a2_scale) v = scaled(v, s0, s1);
                *reinterpret_cast<uint4*>(smem + (q >> 2) * DY_PANEL + px * PIX_ROW + (q & 3) * 16) = v;
            }
            // batch 2: second half of the halo is issued while the first is being stored
#pragma unroll
            for (int it = 0; it < XS - XH; ++it) vd[it] = load_x(XH + it);
#pragma unroll
            for (int it = 0; it < XH; ++it) store_x(it, vx[it]);
#pragma unroll
            for (int it = 0; it < XS - XH; ++it) store_x(XH + it, vd[it]);
        } else {
            // ---- stage dy tile: 256 px x 64 n (16 float4 per pixel), 16 float4 per thread
            {
                const int q = tid & 15;  // float4 index within the 64 channels
                const int nn = n0 + q * 4;
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
                if (p.a2_scale && nn < N) sc = *reinterpret_cast<const float4*>(p.a2_scale + (long)b * N + nn);
    #pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    int px = (tid >> 4) + 16 * it;
                    int y = y0 + px / TW, x = x0 + (px % TW);
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (nn < N && y < H && x < W) v = act_ld4(p.a2, ((long)(b * H + y) * W + x) * N + nn, p.act_bf16);
                    uint2 h;
                    h.x = pack_bf16(v.x * sc.x, v.y * sc.y);
                    h.y = pack_bf16(v.z * sc.z, v.w * sc.w);
                    *reinterpret_cast<uint2*>(smem + (q >> 3) * DY_PANEL + px * PIX_ROW + (q & 7) * 8) = h;
                }
            }
            // ---- stage x halo: 340 px x 64 c
            {
                const int q = tid & 15;
                const int cc = c0 + q * 4;
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
                if (p.a_scale && cc < C) sc = *reinterpret_cast<const float4*>(p.a_scale + (long)b * C + cc);
    #pragma unroll 4
                for (int it = 0; it < (NP + 15) / 16; ++it) {
                    int hp = (tid >> 4) + 16 * it;
                    if (hp < NP) {
                        int hh = hp / HWD, ww = hp - hh * HWD;
                        int y = y0 - 1 + hh, x = x0 - 1 + ww;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (cc < C && y >= 0 && y < H && x >= 0 && x < W)
                            v = act_ld4(p.a, ((long)(b * H + y) * W + x) * C + cc, p.act_bf16);
                        uint2 h;
                        h.x = pack_bf16(v.x * sc.x, v.y * sc.y);
                        h.y = pack_bf16(v.z * sc.z, v.w * sc.w);
                        *reinterpret_cast<uint2*>(smem + X_OFF + (q >> 3) * X_PANEL + hp * PIX_ROW + (q & 7) * 8) = h;
                    }
                }
            }
        }
        __syncthreads();

        // ---- 16 k-steps of 16 pixels (half a tile row each); 9 taps share the A operand.  The tap set is a
        // compile-time mask (space-to-depth tiles dispatch on their 4 possible masks) so that a k-step is
        // straight-line code: all its LDS transpose reads are issued first and the MFMAs start as operands
        // arrive (a per-tap runtime branch made every MFMA wait for its own read: ~160 cycles per 32-cycle MFMA).
        const char* a_base = smem + wn * DY_PANEL + lane_off;
        const char* b_base = smem + X_OFF + wc * X_PANEL + lane_off;
        if (!S2D) compute_tile<TW, 0x1ffu>(acc, a_base, b_base);
        else if (tapmask == 0x010u) compute_tile<TW, 0x010u>(acc, a_base, b_base);
        else if (tapmask == 0x018u) compute_tile<TW, 0x018u>(acc, a_base, b_base);
        else if (tapmask == 0x012u) compute_tile<TW, 0x012u>(acc, a_base, b_base);
        else compute_tile<TW, 0x01bu>(acc, a_base, b_base);
        __syncthreads();
    }

    // ---- partial[split][n][tap][c]; D[i=n][j=c]: col j = lane&31, row i = (r&3)+8*(r>>2)+4*(lane>>5)
    const int lj = lane & 31, lh = lane >> 5;
    float* out = p.y + (long)split * N * 9 * C;
    const int c = c0 + wc * 32 + lj;
    if (c < C) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < N) out[((long)n * 9 + t) * C + c] = acc[t][r];
            }
    }
}

// ---- LDS-DMA variant for the memory-latency-bound low-channel layers ---------------------------------------
// PMC on the 64-channel 256 px layers: 54 % of the wave cycles of the kernel above are spent waiting for its
// staging loads (144 accumulator VGPRs leave no room to prefetch the next tile in registers, 75 KB of LDS per
// block none for a second buffer at 2 blocks/CU).  Here ONE block per CU owns both 75 KB buffers and stages with
// global_load_lds_dwordx4 (global -> LDS, no VGPRs): tile t+1 is in flight while tile t is being multiplied, so a
// CU always has 76 KB outstanding.  A wave instruction writes 1 KiB of LDS linearly (lane l -> +16 l bytes) =
// 16 pixel rows x 64 B of one 32-channel panel — exactly the unpadded panel layout the transpose reads want —
// while the per-lane GLOBAL address is free, so halo / image-border pixels point at a 16-byte zero page.
// bf16 activations, no per-sample scales (discriminator / encoder layers), TW = 32.
__device__ uint4 g_zero_page[4];

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gl_void_ptr;

struct GeomDma {
    static constexpr int NPR = 352;                       // 340 halo pixels rounded up to 16-pixel DMA pieces
    static constexpr int DY_PANEL = 256 * PIX_ROW + 64;
    static constexpr int X_PANEL = NPR * PIX_ROW + 64;
    static constexpr int X_OFF = 2 * DY_PANEL;
    static constexpr int BUF = 2 * DY_PANEL + 2 * X_PANEL;
    static constexpr int SMEM_BYTES = 2 * BUF;            // 152.5 KiB: one block per CU
};

// WAVES = 8 (plain layers): two wave quartets share the staged tiles and split the 9 taps (0-4 / 5-8), i.e. TWO waves
// per SIMD.  With one wave per SIMD the transpose reads (ds_read_b64_tr_b16, 10 per MFMA group) ran far below the LDS
// rate — 8-byte LDS reads need several waves per SIMD in flight — and each wave's MFMAs waited for its own reads: per
// tile 4608 MFMA cycles + ~4750 cycles of DMA + the read stalls came to ~10300 cycles, fully serialised.
template <bool S2D, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, 1) void conv3x3_wgrad_halo_dma_kernel(ConvKParams p) {
    constexpr int TW = 32, TH = 8, HWD = 34, NP = 340;
    constexpr int DY_PANEL = GeomDma::DY_PANEL, X_PANEL = GeomDma::X_PANEL, X_OFF = GeomDma::X_OFF, BUF = GeomDma::BUF;
    static_assert(DY_PANEL == Geom<32>::DY_PANEL, "compute_tile addresses the dy panels with Geom<32>'s pitch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(WAVES == 4 || (WAVES == 8 && !S2D), "the tap split is for the plain 9-tap layers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = (wave & 3) >> 1, wc = wave & 1;
    const int tg = wave >> 2;  // tap group of this wave (WAVES == 8): 0 -> taps 0..4, 1 -> taps 5..8
    const int H = p.Ho, W = p.Wo, C = p.Ck, N = p.N;
    const int n_tiles = (N + 63) / 64, c_tiles = (C + 63) / 64;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles_img = tiles_x * tiles_y;
    const int total_tiles = p.B * tiles_img;
    int bid = blockIdx.x;
    {
        int nblk = gridDim.x, q = nblk >> 3, rr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int ot = bid % (n_tiles * c_tiles);
    const int split = bid / (n_tiles * c_tiles);
    const int n0 = (ot / c_tiles) * 64, c0 = (ot % c_tiles) * 64;
    const int t_begin = split * (int)p.split_len;
    const int t_end = min(total_tiles, t_begin + (int)p.split_len);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const unsigned tapmask = S2D ? stylex_s2d_tap_mask((c0 + wc * 32) / p.s2d_c) : 0x1ffu;
    const int i16 = lane & 15, g = lane >> 4;
    const int lane_off = ((g >> 1) * 8 + (i16 >> 2)) * PIX_ROW + ((g & 1) * 16 + (i16 & 3) * 4) * 2;
    // bias gradient: the 4-tap quartet of the blocks of channel tile 0 (one of its two column waves) carries it
    const bool do_bias = WAVES == 8 && p.bias_partial != nullptr && tg == 1 && c0 == 0 && wc == 0;

    const unsigned short* dy = reinterpret_cast<const unsigned short*>(p.a2);
    const unsigned short* xs = reinterpret_cast<const unsigned short*>(p.a);
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero_page);
    const int lp = lane >> 2, slot = lane & 3;  // pixel row within a 16-pixel piece, 16-byte slot within its 64 B

    auto issue = [&](int tile, int buf) {
        const int b = tile / tiles_img;
        const int tt = tile - b * tiles_img;
        const int y0 = (tt / tiles_x) * TH, x0 = (tt % tiles_x) * TW;
        char* base = smem + buf * BUF;
#pragma unroll
        for (int it = 0; it < 32 / WAVES; ++it) {  // dy: 2 panels x 16 pieces of 16 pixels
            const int ch = wave + WAVES * it, panel = ch >> 4, piece = ch & 15;
            const int px = piece * 16 + lp;
            const int y = y0 + (px >> 5), x = x0 + (px & 31);
            const int nn = n0 + panel * 32 + slot * 8;
            const unsigned short* src = (nn < N && y < H && x < W) ? dy + ((long)(b * H + y) * W + x) * N + nn : zero;
            __builtin_amdgcn_global_load_lds((gl_void_ptr)src, (lds_void_ptr)(base + panel * DY_PANEL + piece * 16 * PIX_ROW),
                                             16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < (44 + WAVES - 1) / WAVES; ++it) {  // x halo: 2 panels x 22 pieces
            const int ch = wave + WAVES * it;
            if (ch < 44) {
                const int panel = ch >= 22 ? 1 : 0, piece = ch - panel * 22;
                const int hp = piece * 16 + lp;
                const int hh = hp / HWD, ww = hp - hh * HWD;
                const int y = y0 - 1 + hh, x = x0 - 1 + ww;
                const int cc = c0 + panel * 32 + slot * 8;
                const bool ok = hp < NP && cc < C && y >= 0 && y < H && x >= 0 && x < W;
                const unsigned short* src = ok ? xs + ((long)(b * H + y) * W + x) * C + cc : zero;
                __builtin_amdgcn_global_load_lds((gl_void_ptr)src,
                                                 (lds_void_ptr)(base + X_OFF + panel * X_PANEL + piece * 16 * PIX_ROW), 16, 0, 0);
            }
        }
    };

    if (t_begin < t_end) issue(t_begin, 0);
    int buf = 0;
    for (int tile = t_begin; tile < t_end; ++tile, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of `tile` have landed
        __syncthreads();                                   // ... and everybody else's; compute(tile-1) is finished
        if (tile + 1 < t_end) issue(tile + 1, buf ^ 1);
        const char* a_base = smem + buf * BUF + wn * DY_PANEL + lane_off;
        const char* b_base = smem + buf * BUF + X_OFF + wc * X_PANEL + lane_off;
        if (WAVES == 8) {
            if (tg == 0) compute_tile<TW, 0x01fu>(acc, a_base, b_base);
            else if (do_bias) compute_tile<TW, 0x1e0u, true>(acc, a_base, b_base);
            else compute_tile<TW, 0x1e0u>(acc, a_base, b_base);
        } else if (!S2D) compute_tile<TW, 0x1ffu>(acc, a_base, b_base);
        else if (tapmask == 0x010u) compute_tile<TW, 0x010u>(acc, a_base, b_base);
        else if (tapmask == 0x018u) compute_tile<TW, 0x018u>(acc, a_base, b_base);
        else if (tapmask == 0x012u) compute_tile<TW, 0x012u>(acc, a_base, b_base);
        else compute_tile<TW, 0x01bu>(acc, a_base, b_base);
    }

    const int lj = lane & 31, lh = lane >> 5;
    float* out = p.y + (long)split * N * 9 * C;
    const int c = c0 + wc * 32 + lj;
    if (do_bias && lj == 0) {  // every column of acc[0] holds the same sums: column 0 writes them
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (n < N) p.bias_partial[(long)split * N + n] = acc[0][r];
        }
    }
    if (c < C) {
        // modulated layer: the whole split lies in sample t_begin / tiles_img (plan), its x scale is a factor of the sum
        const float xsc = (p.a_scale && t_begin < t_end) ? p.a_scale[(long)(t_begin / tiles_img) * C + c] : 1.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (WAVES == 8 && (t < 5) != (tg == 0)) continue;  // the other quartet's taps
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < N) out[((long)n * 9 + t) * C + c] = acc[t][r] * xsc;
            }
        }
    }
}

}  // namespace

bool stylex_wgrad_halo_applicable(const ConvKParams& p) {
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1) return false;
    if (p.Hi != p.Ho || p.Wi != p.Wo) return false;
    if (p.Ck % 4 != 0 || p.N % 4 != 0 || p.Wo < 16 || p.Ho < 8 || (long)p.Ho * p.Wo < 256) return false;
    if (p.Ck == 8 && p.act_bf16 && p.N % 8 == 0) return false;  // RGB input: the flattened-tap kernel (conv_wgrad_tr.hip)
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.a2) & 15)) return false;
    if (p.a_scale && (reinterpret_cast<uintptr_t>(p.a_scale) & 15)) return false;
    if (p.a2_scale && (reinterpret_cast<uintptr_t>(p.a2_scale) & 15)) return false;
    return true;
}

static void tile_dims(const ConvKParams& p, int* tw, int* th) {
    *tw = p.Wo >= 32 ? 32 : 16;
    *th = 256 / *tw;
}

// the LDS-DMA variant: bf16, unscaled, 32-wide tiles, few output tiles (<= 128 channels on both sides: the layers
// whose time is staging latency, not MFMA)
static bool wgrad_dma_eligible(const ConvKParams& p) {
    static const bool off = getenv("STYLEX_WGRAD_DMA") && getenv("STYLEX_WGRAD_DMA")[0] == '0';
    if (off) return false;
    // a per-sample scale of x (the modulation s[b][c] of a generator layer) is applied to the ACCUMULATORS when a split
    // ends (dW = sum_b s[b][c] * sum_p dy x: linear, and a split never crosses a sample — see the plan); a scale of dy
    // cannot be factored that way (the modulated backward hands in dy with the demodulation already folded in)
    if (!p.act_bf16 || p.Ck % 8 != 0 || p.N % 8 != 0 || p.Wo < 32 || p.a2_scale) return false;
    if (p.a_scale && (reinterpret_cast<uintptr_t>(p.a_scale) & 3)) return false;
    if (p.s2d_c) return false;  // space-to-depth tiles do 1-4 taps per staged tile: 2 blocks per CU hide that better
    static const long max_otiles = getenv("STYLEX_WGRAD_DMA_OT") ? atol(getenv("STYLEX_WGRAD_DMA_OT")) : 16;
    long otiles = (long)((p.N + 63) / 64) * ((p.Ck + 63) / 64);
    return otiles <= max_otiles;
}

void stylex_wgrad_halo_plan(const ConvKParams& p, int* splits, int* tiles_per_split) {
    int tw, th;
    tile_dims(p, &tw, &th);
    long tiles = (long)p.B * ((p.Wo + tw - 1) / tw) * ((p.Ho + th - 1) / th);
    long otiles = (long)((p.N + 63) / 64) * ((p.Ck + 63) / 64);
    if (wgrad_dma_eligible(p)) {  // one resident block per CU
        long want = (256 + otiles - 1) / otiles;
        if (want > tiles) want = tiles;
        if (want < 1) want = 1;
        long tps = (tiles + want - 1) / want;
        if (p.a_scale) {  // splits must not cross a sample: the largest divisor of tiles-per-image that is <= tps
            const long tiles_img = tiles / p.B;
            if (tps > tiles_img) tps = tiles_img;
            while (tiles_img % tps) --tps;
        }
        *tiles_per_split = (int)tps;
        *splits = (int)((tiles + tps - 1) / tps);
        return;
    }
    // ~2 resident blocks per CU (measured: halving the splits of the 512x512 layers to save partial traffic
    // costs 40-60 % in kernel time — parallelism matters more)
    long want = (512 + otiles - 1) / otiles;
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    long tps = (tiles + want - 1) / want;
    *tiles_per_split = (int)tps;
    *splits = (int)((tiles + tps - 1) / tps);
}

template <int TW, bool ABF, bool S2D>
static int launch_wgrad_halo(const ConvKParams& p, int blocks, hipStream_t s) {
    auto k = conv3x3_wgrad_halo_kernel<TW, ABF, S2D>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Geom<TW>::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    stylex_note_kernel("conv3x3_wgrad_halo_kernel<%d, %s, %s>", TW, ABF ? "true" : "false", S2D ? "true" : "false");
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), Geom<TW>::SMEM_BYTES, s, p);
    return (int)hipGetLastError();
}

int stylex_launch_wgrad_halo(ConvKParams p, float* partial, hipStream_t s, int* splits_out, int* bias_done) {
    int splits, tps;
    stylex_wgrad_halo_plan(p, &splits, &tps);
    p.split_len = tps;
    p.y = partial;
    if (bias_done) *bias_done = 0;
    int blocks = ((p.N + 63) / 64) * ((p.Ck + 63) / 64) * splits;
    *splits_out = splits;
    if (wgrad_dma_eligible(p)) {
        static const bool w8 = !(getenv("STYLEX_WGRAD_DMA_W8") && getenv("STYLEX_WGRAD_DMA_W8")[0] == '0');
        const int v = p.s2d_c ? 1 : (w8 ? 2 : 0);
        if (v != 2) p.bias_partial = nullptr;
        else if (p.bias_partial && bias_done) *bias_done = 1;
        auto k = v == 1 ? conv3x3_wgrad_halo_dma_kernel<true, 4> : v == 2 ? conv3x3_wgrad_halo_dma_kernel<false, 8>
                                                                          : conv3x3_wgrad_halo_dma_kernel<false, 4>;
        static bool attr_done[3] = {false, false, false};
        if (!attr_done[v]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               GeomDma::SMEM_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_done[v] = true;
        }
        stylex_note_kernel("conv3x3_wgrad_halo_dma_kernel<%s, %d>", v == 1 ? "true" : "false", v == 2 ? 8 : 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(v == 2 ? 512 : 256), GeomDma::SMEM_BYTES, s, p);
        return (int)hipGetLastError();
    }
    const bool abf = p.act_bf16 && p.Ck % 8 == 0 && p.N % 8 == 0;
    if (p.s2d_c) {
        if (abf) return p.Wo >= 32 ? launch_wgrad_halo<32, true, true>(p, blocks, s) : launch_wgrad_halo<16, true, true>(p, blocks, s);
        return p.Wo >= 32 ? launch_wgrad_halo<32, false, true>(p, blocks, s) : launch_wgrad_halo<16, false, true>(p, blocks, s);
    }
    if (abf) return p.Wo >= 32 ? launch_wgrad_halo<32, true, false>(p, blocks, s) : launch_wgrad_halo<16, true, false>(p, blocks, s);
    return p.Wo >= 32 ? launch_wgrad_halo<32, false, false>(p, blocks, s) : launch_wgrad_halo<16, false, false>(p, blocks, s);
}
