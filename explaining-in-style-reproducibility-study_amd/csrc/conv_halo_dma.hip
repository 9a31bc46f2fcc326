// conv_halo_dma.hip — 3x3 / stride-1 / pad-1 forward and data gradient for the >= 128-channel layers of the
// discriminator / encoder (bf16 activations, no per-sample scales), staged with LDS-DMA.
//
// Why a second halo kernel.  conv_halo.hip stages through registers (60 VGPRs of prefetch + 64 of accumulators at 2
// blocks per CU = the whole 256-register budget) and reads 1 KB of LDS per MFMA (64 px x 64 n wave tiles) — exactly
// the LDS port limit, measured 42 % of the MFMA peak at best.  global_load_lds_dwordx4 needs no staging registers, so
// one block per CU can afford a 16x32-pixel x 128-channel tile (wave tile 128 px x 128 n = 4x4 MFMA tiles, 256
// accumulator registers): every operand read feeds four MFMAs (0.5 KB of LDS per MFMA), the weights of a chunk are
// staged once per 512 pixels instead of once per 256, and the next 16-channel chunk streams into the second LDS
// buffer while this one is multiplied.
//
// LDS image of a chunk (16 input channels = 32-byte rows, unpadded — 8 consecutive rows cover the 256-byte bank
// period once): halo rows [18*34 -> 640][32 B], then weight rows [9 taps][128 n][32 B].  A DMA wave instruction writes
// 1 KiB linearly (lane l -> +16 l bytes) = 32 rows; the per-lane global address is free, so padding pixels and
// channel tails read a 16-byte zero page.
//
// Epilogue: bias + (Leaky)ReLU, the activation gate of a data gradient, the residual merge of the space-to-depth forward
// — applied to the accumulators and stored from registers (see below); the data gradient passes flip_taps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Epilogue without LDS: the MFMA operands are swapped (D^T = W x X^T), so a lane's accumulators are 4 CONSECUTIVE
// CHANNELS of one pixel; one v_permlane32_swap per dword pairs the two half-waves' quads into 8 consecutive channels and
// the lane stores 16 bytes straight from registers (the LDS transpose it replaces cost 128 ds_write_b16 + 16
// ds_read_b128 per wave and tile: 64->64 @256^2 .393 -> .375 ms forward, .390 -> .374 data gradient at B=64).
namespace {

__device__ uint4 g_zero_page_fwd[4];

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gl_void_ptr;

__device__ __forceinline__ unsigned short to_bf16(float v) {
    f32x2_t t = {v, 0.f};
    bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
    return (unsigned short)(*reinterpret_cast<unsigned*>(&r) & 0xffffu);
}

constexpr int TW = 32, TH = 16, HWD = TW + 2, NP = (TH + 2) * HWD;  // 612 halo pixels
constexpr int ROW = 32;                                              // bytes per LDS row (16 bf16)
constexpr int HALO_PIECES = (NP + 31) / 32;                          // 20 DMA pieces of 32 rows
constexpr int HALO_BYTES = HALO_PIECES * 1024;
// TNJ = 32-channel output sub-tiles per wave: 4 -> 128-channel block tile, 256 accumulator registers, 112 KiB of LDS,
// one block per CU;  2 -> 64-channel tile for the N = 64 layers (64->64 @256^2, the 128->64 data gradient): 128
// accumulator registers and 76 KiB, two blocks per CU, so one block's prologue / epilogue hides under the other's
// MFMAs (the register-staged kernel spends ~2000 issue slots per wave and tile on staging and epilogue there).
template <int TNJ>
struct DmaCfg {
    static constexpr int BN = TNJ * 32;
    static constexpr int W_PIECES = 9 * BN / 32;
    static constexpr int PIECES = HALO_PIECES + W_PIECES;      // 56 / 38
    static constexpr int NIT = (PIECES + 3) / 4;               // DMA pieces per wave and chunk: 14 / 10
    static constexpr int BUF = PIECES * 1024;                  // 57344 / 38912
    static constexpr int SMEM_BYTES = 2 * BUF;
};

// One 16-channel chunk of MFMAs for the spatial taps in MASK; the chunk's weights are staged compactly, slot = rank of
// the tap within MASK (space-to-depth data gradient: 1, 2 or 4 of the 9 taps exist for an output sub-position).
template <int TNJ, unsigned MASK>
__device__ __forceinline__ void dma_chunk_masked(const char* base, const int (&a_off)[6][3], int b_lane,
                                                 f32x16 (&acc)[4][TNJ]) {
    constexpr int BN = TNJ * 32;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (!((MASK >> tap) & 1u)) continue;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int slot = __builtin_popcount(MASK & ((1u << tap) - 1u));
        bf16x8 av[4], bv[TNJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const bf16x8*>(base + a_off[i + kh][kw]);
#pragma unroll
        for (int j = 0; j < TNJ; ++j)
            bv[j] = *reinterpret_cast<const bf16x8*>(base + b_lane + (slot * BN + j * 32) * ROW);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TNJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j], av[i], acc[i][j], 0, 0, 0);  // D^T: rows = channels
    }
}

// spatial taps of the space-to-depth DATA GRADIENT per output sub-position q = (sy, sx): weight tap 8 - t must be one
// of the structurally non-zero taps of stylex_s2d_tap_mask(q)
constexpr unsigned S2D_DGRAD_MASK[4] = {1u << 4, (1u << 4) | (1u << 5), (1u << 4) | (1u << 7),
                                        (1u << 4) | (1u << 5) | (1u << 7) | (1u << 8)};

// FORWARD of the space-to-depth conv: the tap set depends on the sub-position q of the INPUT-channel chunk
// (stylex_s2d_tap_mask(q)); taps listed low to high as nibbles for the staging slot -> tap lookup.
constexpr unsigned S2D_FWD_MASK[4] = {1u << 4, (1u << 3) | (1u << 4), (1u << 1) | (1u << 4),
                                      (1u << 0) | (1u << 1) | (1u << 3) | (1u << 4)};
__device__ __forceinline__ int s2d_fwd_taps(int q) { return q == 0 ? 0x4 : q == 1 ? 0x43 : q == 2 ? 0x41 : 0x4310; }
__device__ __forceinline__ int s2d_fwd_ntaps(int q) { return q == 0 ? 1 : q == 3 ? 4 : 2; }

template <int TNJ, bool S2D = false>
__global__ __launch_bounds__(256, TNJ == 4 ? 1 : 2) void conv3x3_halo_dma_kernel(ConvKParams p) {
    constexpr int BN = DmaCfg<TNJ>::BN, PIECES = DmaCfg<TNJ>::PIECES, NIT = DmaCfg<TNJ>::NIT, BUF = DmaCfg<TNJ>::BUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Ho, W = p.Wo, C = p.Ck, N = p.N;
    const int n_tiles = (N + BN - 1) / BN;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;

    int bid = blockIdx.x;
    {  // XCD-aware order: the output-channel tiles of one pixel tile are neighbours on one XCD
        int nblk = gridDim.x, q = nblk >> 3, rr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int n0 = (bid % n_tiles) * BN;
    int pt = bid / n_tiles;
    const int b = pt / (tiles_x * tiles_y);
    pt -= b * tiles_x * tiles_y;
    const int y0 = (pt / tiles_x) * TH, x0 = (pt % tiles_x) * TW;

    const unsigned short* xs = reinterpret_cast<const unsigned short*>(p.a);
    const unsigned short* ws = reinterpret_cast<const unsigned short*>(p.w);
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero_page_fwd);
    const int lr = lane >> 1, pslot = lane & 1;  // row within a 32-row piece, PHYSICAL 16-byte half of the 32-byte row
    // Bank swizzle: LDS row R keeps channel half h at physical half h ^ bit3(R).  An MFMA operand read touches 16
    // consecutive rows x ONE half per 16-lane pass (lane = row + 32 * half): unswizzled, rows r and r + 8 collide
    // (32-byte rows = 8 banks, 64 banks) — measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.48; with the swap
    // the 16 rows cover all 64 banks once.  Every piece starts at a multiple of 32 rows, so bit3(R) = bit3(lr).
    const int slot = pslot ^ ((lr >> 3) & 1);   // logical channel half this lane fetches

    // per-wave piece list (same for every chunk): pieces wave, wave+4, ...; global element offsets without the chunk term
    long src_off[NIT];
    bool src_ok[NIT];
    int wslot[NIT];  // weight pieces: tap slot of the row this lane stages (S2D forward)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int piece = wave + 4 * it;
        src_off[it] = 0;
        src_ok[it] = false;
        wslot[it] = 0;
        if (piece >= PIECES) continue;  // 38 pieces: waves 2 and 3 have nine
        if (piece < HALO_PIECES) {
            const int hp = piece * 32 + lr;
            const int hh = hp / HWD, ww = hp - hh * HWD;
            const int y = y0 - 1 + hh, x = x0 - 1 + ww;
            src_ok[it] = hp < NP && y >= 0 && y < H && x >= 0 && x < W;
            src_off[it] = ((long)(b * H + y) * W + x) * C + slot * 8;
        } else {
            const int r = (piece - HALO_PIECES) * 32 + lr;  // weight row = tap (S2D: tap slot) * BN + n
            int tap = r / BN;
            const int nl = r - tap * BN;
            bool tap_ok = true;
            wslot[it] = tap;
            if (S2D && !p.flip_taps) {  // forward: the tap of a slot changes with the chunk's sub-position (added at issue time)
                tap = 0;
            } else if (S2D) {  // data gradient: slot -> the slot-th spatial tap of this block's sub-position
                const unsigned m = S2D_DGRAD_MASK[n0 / p.s2d_c];
                tap_ok = tap < __builtin_popcount(m);
                unsigned mm = m;
                for (int k = 0; k < tap; ++k) mm &= mm - 1;  // drop the lowest set bits
                tap = tap_ok ? __builtin_ctz(mm) : 0;
            }
            const int gt = p.flip_taps ? 8 - tap : tap;
            src_ok[it] = tap_ok && n0 + nl < N;
            src_off[it] = ((long)(n0 + nl) * 9 + gt) * C + slot * 8;
        }
    }

    auto issue_piece = [&](int c0, int buf, int it) {
        char* base = smem + buf * BUF;
        const bool cok = c0 + slot * 8 < C;
        const int piece = wave + 4 * it;
        if (piece >= PIECES) return;
        if (S2D && piece >= HALO_PIECES + 4 * TNJ) return;  // at most 4 tap slots are ever staged
        long off = src_off[it] + c0;
        bool ok = src_ok[it] && cok;
        if (S2D && !p.flip_taps && piece >= HALO_PIECES) {
            const int qn = c0 / p.s2d_c;
            ok = ok && wslot[it] < s2d_fwd_ntaps(qn);
            off += (long)((s2d_fwd_taps(qn) >> (4 * wslot[it])) & 0xF) * C;
        }
        const unsigned short* g = (piece < HALO_PIECES ? xs : ws) + off;
        const unsigned short* src = ok ? g : zero;
        __builtin_amdgcn_global_load_lds((gl_void_ptr)src, (lds_void_ptr)(base + piece * 1024), 16, 0, 0);
    };
    auto issue = [&](int c0, int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) issue_piece(c0, buf, it);
    };
    // Second K segment of the space-to-depth FORWARD (its chunks come FIRST and share the centre-tap loop of sub-position
    // 0): 16 channels [cx0, cx0 + 16) of the residual operand x2 [B][H][W][c2], weights w2 [N][c2].  Kept out of
    // issue_piece — inside it the extra code stopped the per-piece arrays from living in registers (640 bytes of scratch,
    // the step 27 % slower) — and recomputed per piece: a handful of chunks per tile.
    auto issue_extra = [&](int cx0, int buf) {
        char* base = smem + buf * BUF;
        const unsigned short* x2 = reinterpret_cast<const unsigned short*>(p.x2);
        const unsigned short* w2 = reinterpret_cast<const unsigned short*>(p.w2);
        const int cx = cx0 + slot * 8;
        for (int piece = wave; piece < HALO_PIECES + TNJ; piece += 4) {  // halo pieces + ONE tap slot of weights (BN rows)
            const unsigned short* src = zero;
            if (piece < HALO_PIECES) {
                const int hp = piece * 32 + lr;
                const int hh = hp / HWD, ww = hp - hh * HWD;
                const int y = y0 - 1 + hh, x = x0 - 1 + ww;
                if (hp < NP && y >= 0 && y < H && x >= 0 && x < W && cx < p.c2) src = x2 + ((long)(b * H + y) * W + x) * p.c2 + cx;
            } else {
                const int nl = (piece - HALO_PIECES) * 32 + lr;
                if (n0 + nl < N && cx < p.c2) src = w2 + (long)(n0 + nl) * p.c2 + cx;
            }
            __builtin_amdgcn_global_load_lds((gl_void_ptr)src, (lds_void_ptr)(base + piece * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[4][TNJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TNJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // operand addressing: MFMA row-tile i of this wave = tile row 4*wave + i (32 pixels); lane li <-> pixel / channel
    const int li = lane & 31, lk = lane >> 5;
    const int a_row = (4 * wave) * HWD + li;  // halo row of this lane's pixel for (i, kh, kw) = (0, 0, 0)
    int a_off[6][3];                          // swizzled byte offsets of the 18 distinct (i + kh, kw) operand rows
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int t = a_row + r * HWD + kw;
            a_off[r][kw] = (t * ROW) | ((((t >> 3) ^ lk) & 1) << 4);
        }
    // weight rows tap*BN + j*32 + li: the tap / j terms are multiples of 16 rows, so bit3(row) = bit3(li)
    const int b_lane = HALO_BYTES + li * ROW + ((((li >> 3) ^ lk) & 1) << 4);  // + (tap*128 + j*32)*ROW

    const int chunks2 = (S2D && !p.flip_taps && p.x2) ? (p.c2 + 15) / 16 : 0;  // second K segment (residual 1x1 conv)
    const int nchunks = (C + 15) / 16 + chunks2;
    if (chunks2) issue_extra(0, 0);
    else issue(0, 0);
    int buf = 0;
    if (S2D) {
        // the tap set is uniform per block (output sub-position of its channel tile): one copy of the whole chunk loop
        // per set (a switch inside the loop made the 128 accumulator registers a four-way phi and spilled them)
        auto run = [&](auto mask_tag, int ch_begin, int ch_end) {
            constexpr unsigned MASK = decltype(mask_tag)::value;
            for (int ch = ch_begin; ch < ch_end; ++ch, buf ^= 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (ch + 1 < chunks2) issue_extra((ch + 1) * 16, buf ^ 1);
                else if (ch + 1 < nchunks) issue((ch + 1 - chunks2) * 16, buf ^ 1);
                dma_chunk_masked<TNJ, MASK>(smem + buf * BUF, a_off, b_lane, acc);
            }
        };
        if (p.flip_taps) {
            switch (n0 / p.s2d_c) {
                case 0: run(std::integral_constant<unsigned, S2D_DGRAD_MASK[0]>{}, 0, nchunks); break;
                case 1: run(std::integral_constant<unsigned, S2D_DGRAD_MASK[1]>{}, 0, nchunks); break;
                case 2: run(std::integral_constant<unsigned, S2D_DGRAD_MASK[2]>{}, 0, nchunks); break;
                default: run(std::integral_constant<unsigned, S2D_DGRAD_MASK[3]>{}, 0, nchunks); break;
            }
        } else {  // forward: the four input sub-positions one after the other, each with its own tap set
            const int cpq = p.s2d_c / 16, c2 = chunks2;  // the residual segment's chunks lead: centre tap, like sub-position 0
            run(std::integral_constant<unsigned, S2D_FWD_MASK[0]>{}, 0, c2 + cpq);
            run(std::integral_constant<unsigned, S2D_FWD_MASK[1]>{}, c2 + cpq, c2 + 2 * cpq);
            run(std::integral_constant<unsigned, S2D_FWD_MASK[2]>{}, c2 + 2 * cpq, c2 + 3 * cpq);
            run(std::integral_constant<unsigned, S2D_FWD_MASK[3]>{}, c2 + 3 * cpq, c2 + 4 * cpq);
        }
    } else
    for (int ch = 0; ch < nchunks; ++ch, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = ch + 1 < nchunks;
        const char* base = smem + buf * BUF;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            bf16x8 av[4], bv[TNJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                av[i] = *reinterpret_cast<const bf16x8*>(base + a_off[i + kh][kw]);
#pragma unroll
            for (int j = 0; j < TNJ; ++j)
                bv[j] = *reinterpret_cast<const bf16x8*>(base + b_lane + (tap * BN + j * 32) * ROW);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TNJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j], av[i], acc[i][j], 0, 0, 0);  // D^T: rows = channels
            // the next chunk's 14 DMA pieces are issued two per tap, in the shadow of this tap's 16 MFMAs (a burst at
            // the top of the chunk leaves the MFMA pipe empty while it is issued)
            if (more && 2 * tap < NIT) {
                issue_piece((ch + 1) * 16, buf ^ 1, 2 * tap);
                if (2 * tap + 1 < NIT) issue_piece((ch + 1) * 16, buf ^ 1, 2 * tap + 1);
            }
        }
    }
    // ---- epilogue, straight from the accumulators.  D[row = channel][col = pixel]: lane (lj = pixel of the row-tile, lh)
    // holds channels j*32 + 8g + 4lh + (0..3) for g = r >> 2.
    const int lj = lane & 31, lh = lane >> 5;
    const bool act = (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) != 0;
    const float slope = (p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f;
    unsigned short* yout = reinterpret_cast<unsigned short*>(p.y);
    const unsigned short* gate = (!S2D && (p.flags & STYLEX_EPI_GATE)) ? reinterpret_cast<const unsigned short*>(p.residual) : nullptr;
    const float gslope = p.res_scale;
    const unsigned char* gmask = S2D ? nullptr : p.gate_mask;  // STYLEX_EPI_GATE_MASK: the gate as one bit per element
    unsigned char* mask_out = S2D ? nullptr : p.mask;          // STYLEX_EPI_MASK_OUT
    auto gatem = [&](unsigned u, unsigned bits) -> unsigned {  // two bf16 of dx, gate bits (element > 0) in bits 0 and 1
        const float a0 = __uint_as_float(u << 16), c0 = __uint_as_float(u & 0xffff0000u);
        return (unsigned)to_bf16((bits & 1u) ? a0 : gslope * a0) | ((unsigned)to_bf16((bits & 2u) ? c0 : gslope * c0) << 16);
    };
    auto gate2 = [&](unsigned u, unsigned g) -> unsigned {  // two bf16 of dx times the LeakyReLU derivative at the gate
        const float a0 = __uint_as_float(u << 16), c0 = __uint_as_float(u & 0xffff0000u);
        const float ga = __uint_as_float(g << 16), gc = __uint_as_float(g & 0xffff0000u);
        return (unsigned)to_bf16(ga > 0.f ? a0 : gslope * a0) | ((unsigned)to_bf16(gc > 0.f ? c0 : gslope * c0) << 16);
    };
    {
        // block merge of DiscriminatorBlock (:743) on the space-to-depth forward: (conv + bias + residual) * res_scale in fp32
        const unsigned short* res = (S2D && (p.flags & STYLEX_EPI_RESIDUAL)) ? reinterpret_cast<const unsigned short*>(p.residual) : nullptr;
        const float rsc = p.res_scale;
        const bool merged = S2D && !p.flip_taps && p.x2 != nullptr;  // the residual conv sits in the accumulators already
        // D[row = channel][col = pixel]: lane (lj = pixel, lh) holds channels j*32 + 8g + 4lh + (0..3), g = r >> 2
        float4 b4[TNJ][4];
#pragma unroll
        for (int j = 0; j < TNJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + j * 32 + 8 * g + 4 * lh;
                b4[j][g] = ((p.flags & STYLEX_EPI_BIAS) && n < N) ? *reinterpret_cast<const float4*>(p.bias + n)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 + 4 * wave + i, x = x0 + lj;
            const bool pix_ok = y < H && x < W;
            const long obase = ((long)(b * H + y) * W + x) * N;
#pragma unroll
            for (int j = 0; j < TNJ; ++j) {
                unsigned P[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[i][j][4 * g + 0] + b4[j][g].x, v1 = acc[i][j][4 * g + 1] + b4[j][g].y;
                    float v2 = acc[i][j][4 * g + 2] + b4[j][g].z, v3 = acc[i][j][4 * g + 3] + b4[j][g].w;
                    if (res) {
                        const int n = n0 + j * 32 + 8 * g + 4 * lh;
                        uint2 rv = make_uint2(0u, 0u);
                        if (pix_ok && n < N) rv = *reinterpret_cast<const uint2*>(res + obase + n);
                        v0 = (v0 + __uint_as_float(rv.x << 16)) * rsc;
                        v1 = (v1 + __uint_as_float(rv.x & 0xffff0000u)) * rsc;
                        v2 = (v2 + __uint_as_float(rv.y << 16)) * rsc;
                        v3 = (v3 + __uint_as_float(rv.y & 0xffff0000u)) * rsc;
                    } else if (merged) {
                        v0 *= rsc;
                        v1 *= rsc;
                        v2 *= rsc;
                        v3 *= rsc;
                    }
                    if (act) {
                        v0 = v0 > 0.f ? v0 : slope * v0;
                        v1 = v1 > 0.f ? v1 : slope * v1;
                        v2 = v2 > 0.f ? v2 : slope * v2;
                        v3 = v3 > 0.f ? v3 : slope * v3;
                    }
                    P[g][0] = (unsigned)to_bf16(v0) | ((unsigned)to_bf16(v1) << 16);
                    P[g][1] = (unsigned)to_bf16(v2) | ((unsigned)to_bf16(v3) << 16);
                }
                // upper half-wave's quad g <-> lower half-wave's quad g+1 (g = 0, 2): afterwards the lower lanes hold channels
                // 8g .. 8g+7 for g = 0, 2 and the upper lanes for g = 1, 3
#pragma unroll
                for (int g = 0; g < 4; g += 2)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        auto r = __builtin_amdgcn_permlane32_swap(P[g][h], P[g + 1][h], false, false);
                        P[g][h] = r[0];
                        P[g + 1][h] = r[1];
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int n = n0 + j * 32 + 16 * q + 8 * lh;
                    if (pix_ok && n < N) {
                        uint4 v = make_uint4(P[2 * q][0], P[2 * q][1], P[2 * q + 1][0], P[2 * q + 1][1]);
                        const long o = obase + n;
                        if (gate) {
                            const uint4 gv = *reinterpret_cast<const uint4*>(gate + o);
                            v.x = gate2(v.x, gv.x);
                            v.y = gate2(v.y, gv.y);
                            v.z = gate2(v.z, gv.z);
                            v.w = gate2(v.w, gv.w);
                        } else if (gmask) {
                            const unsigned m = gmask[o >> 3];
                            v.x = gatem(v.x, m);
                            v.y = gatem(v.y, m >> 2);
                            v.z = gatem(v.z, m >> 4);
                            v.w = gatem(v.w, m >> 6);
                        }
                        *reinterpret_cast<uint4*>(yout + o) = v;
                        if (mask_out) mask_out[o >> 3] = (unsigned char)stylex_sign_bits8(v);
                    }
                }
            }
        }
    }
}

}  // namespace

// STYLEX_NOT_APPLICABLE unless: bf16 activations, plain 3x3/s1/p1, no per-sample scales / noise / residual /
// space-to-depth, whole 8-channel slots, at least one 16x32 tile's worth of image, >= 64 input channels, and an
// output width that is a multiple of 64 (64-channel tiles) or >= 128 with >= 128 input channels (128-channel tiles).
template <int TNJ, bool S2D = false>
static int launch_dma(const ConvKParams& p, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_dma_kernel<TNJ, S2D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DmaCfg<TNJ>::SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    long tiles = (long)p.B * ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    long blocks = tiles * ((p.N + DmaCfg<TNJ>::BN - 1) / DmaCfg<TNJ>::BN);
    stylex_note_kernel("conv3x3_halo_dma_kernel<%d, %s>", TNJ, S2D ? "true" : "false");
    hipLaunchKernelGGL((conv3x3_halo_dma_kernel<TNJ, S2D>), dim3((unsigned)blocks), dim3(256), DmaCfg<TNJ>::SMEM_BYTES, s, p);
    return (int)hipGetLastError();
}

int stylex_launch_halo_dma(const ConvKParams& p, hipStream_t s) {
    static const bool on = !(getenv("STYLEX_HALO_DMA") && getenv("STYLEX_HALO_DMA")[0] == '0');
    static const bool on64 = !(getenv("STYLEX_HALO_DMA64") && getenv("STYLEX_HALO_DMA64")[0] == '0');
    static const bool on_s2d = !(getenv("STYLEX_HALO_DMA_S2D") && getenv("STYLEX_HALO_DMA_S2D")[0] == '0');
    if (!on) return STYLEX_NOT_APPLICABLE;
    if (!p.act_bf16 || p.a_scale) return STYLEX_NOT_APPLICABLE;
    if (p.flags & ~(STYLEX_EPI_BIAS | STYLEX_EPI_LRELU | STYLEX_EPI_RELU |
                    (p.s2d_c ? STYLEX_EPI_RESIDUAL : (STYLEX_EPI_GATE | STYLEX_EPI_GATE_MASK | STYLEX_EPI_MASK_OUT))))
        return STYLEX_NOT_APPLICABLE;
    if (p.s2d_c && (p.mask || p.gate_mask)) return STYLEX_NOT_APPLICABLE;
    if (p.x2) {  // second K segment: forward of the space-to-depth conv only
        if (!p.s2d_c || p.flip_taps || !p.w2 || p.c2 < 8 || p.c2 % 8 != 0 || (p.flags & STYLEX_EPI_RESIDUAL)) return STYLEX_NOT_APPLICABLE;
        if ((reinterpret_cast<uintptr_t>(p.x2) & 15) || (reinterpret_cast<uintptr_t>(p.w2) & 15)) return STYLEX_NOT_APPLICABLE;
    }
    if ((p.flags & STYLEX_EPI_GATE) && (!p.residual || (reinterpret_cast<uintptr_t>(p.residual) & 15))) return STYLEX_NOT_APPLICABLE;
    if (p.s2d_c && on_s2d) {  // round 5: the stride-2 kernels of their own (also 16-pixel-wide images)
        const int rc = p.flip_taps ? stylex_launch_s2d_dgrad(p, s) : stylex_launch_s2d_fwd(p, s);
        if (rc != STYLEX_NOT_APPLICABLE) return rc;
    }
    if (p.N % 8 != 0 || p.Ck % 8 != 0 || p.Wo < 32 || p.Ho < 16) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) ||
        (reinterpret_cast<uintptr_t>(p.y) & 15))
        return STYLEX_NOT_APPLICABLE;
    if (p.s2d_c) {
        // space-to-depth stride-2 conv.  Data gradient: tap set uniform per output-channel tile, no epilogue.
        // Forward: tap set per input sub-position, epilogue bias (+ residual merge).
        if (!on_s2d || !on64 || p.s2d_c % 64 != 0) return STYLEX_NOT_APPLICABLE;
        if (p.flip_taps) {
            if (p.flags || p.N != 4 * p.s2d_c || p.Ck < 64) return STYLEX_NOT_APPLICABLE;
        } else {
            static const bool on_fwd = !(getenv("STYLEX_HALO_DMA_S2D_FWD") && getenv("STYLEX_HALO_DMA_S2D_FWD")[0] == '0');
            if (!on_fwd || (p.flags & ~(STYLEX_EPI_BIAS | STYLEX_EPI_RESIDUAL)) || p.Ck != 4 * p.s2d_c || p.N % 64 != 0)
                return STYLEX_NOT_APPLICABLE;
            if ((p.flags & STYLEX_EPI_RESIDUAL) && (!p.residual || (reinterpret_cast<uintptr_t>(p.residual) & 15)))
                return STYLEX_NOT_APPLICABLE;
        }
        // (round 4: the 128-channel tile <4, true> at one block per CU was measured on these launches — 683 vs 815
        // images/s on the step, profiles/r04_b_ab_s2d_wide.txt: the masked chunk loops with 256 accumulator registers
        // spill — so the 64-channel tile at two blocks per CU stays)
        return p.dry ? 0 : launch_dma<2, true>(p, s);
    }
    // Measured at B = 64 (fwd / dgrad ms, 128-channel tiles at one block per CU -> 64-channel tiles at two):
    // 128->128 @128^2 .361/.351 -> .313/.311, 128->256 @64^2 .176 -> .149, 256->256 @64^2 .284/.281 -> .264/.273,
    // 512->512 @32^2 .259/.264 -> .257/.263, 64->64 @256^2 (register-staged kernel) .430/.425 -> .377/.384: the second
    // resident block hides more than the wider tile saves in LDS reads, so the 64-channel variant is the default and
    // the 128-channel one serves output widths that are not a multiple of 64.
    if (on64 && p.Ck >= 64 && p.N % 64 == 0) return p.dry ? 0 : launch_dma<2>(p, s);
    // narrow outputs (the data gradient of the first conv of D / the encoder: 64 -> 8 padded RGB channels): one
    // 32-channel output sub-tile per wave; HBM-side (the input is 16x the output), the generic kernel ran it at 56 TF/s
    static const bool on32 = !(getenv("STYLEX_HALO_DMA32") && getenv("STYLEX_HALO_DMA32")[0] == '0');
    if (on32 && on64 && p.Ck >= 64 && p.N <= 32) return p.dry ? 0 : launch_dma<1>(p, s);
    if (p.N >= 128 && p.Ck >= 128) return p.dry ? 0 : launch_dma<4>(p, s);
    return STYLEX_NOT_APPLICABLE;
}
