// conv_s2d_dgrad.hip — round 5: data gradient of the 3x3 / stride-2 conv of a DiscriminatorBlock
// (/root/reference/stylex/stylex_train.py:733-736) in its space-to-depth form, ALL FOUR sub-positions per block.
//
//   dx2[b][y][x][s*C + c] = sum_{t in D(s)} sum_n gz[b][y + dy(t)][x + dx(t)][n] * Wb[s*C + c][8 - t][n]
//
// (x2 = the blurred block input stored space-to-depth, [B][H][W][4C]; gz = the gradient of the conv's output, [B][H][W][N];
// D(s) = the 1, 2, 2, 4 taps of sub-position s = (sy, sx) with offsets dy, dx in {0, +1}; Wb = stylex_pack_weight_s2d's
// data-gradient pack [4C][9][N].)
//
// Why a kernel of its own.  conv3x3_halo_dma_kernel<2, true> gives every 64-channel tile of the 4C output channels its own
// block: the gradient halo is staged once per SUB-POSITION for 1-4 taps of MFMA work (8-32 MFMAs per wave between two
// vmcnt(0) + barrier drains; round-4 counters: matrix pipe 24-31 % busy).  Here a block owns a channel group in all four
// sub-positions over a 256-pixel tile (8 x 32, or 16 x 16 for 16-pixel-wide images), so one staged halo feeds all nine
// (sub-position, tap) pairs.  Blocks: 4 waves and a 32-channel group (128 output channels; 75 KB of LDS, two blocks per CU —
// the default) or 8 waves and a 64-channel group (256 output channels, one block per CU: STYLEX_S2D_DGRAD_TILE=1).  A
// persistent block walks a static tile list; the K loop (gz channels, 32 per stage = 19 KB of halo + 18 / 36 KB of the nine
// weight slots) streams through two-deep LDS rings across tile boundaries; the whole DMA of the next stage is issued behind
// the FIRST MFMAs of a stage, one barrier per stage in front of its last MFMAs (conv_wgrad_pipe.hip's recipe).  Vector-memory
// operations complete in issue order, so at a tile's end the DMA of the stage AFTER the next goes out ahead of the tile's 16
// stores and the following barrier waits with vmcnt(16) — everything but the stores.  A wave owns two 32-pixel fragments x
// one 32-channel group x all four sub-positions (acc[2][4] = 128 registers), keeps the six distinct halo fragments of a
// 16-channel k-step in registers and reads three weight fragments per six MFMAs; every wave runs the same instruction stream
// (a per-quartet split of the nine pairs made the accumulators a two-way phi: 705 spilled registers).  LDS rows are 64 bytes
// (32 channels); 16-byte slot q of row R lives at slot q ^ ((R >> 2) & 3), which spreads every lane group of a ds_read_b128
// over all 64 banks.  Epilogue: conv_line64.hip's register transpose; a store writes the 64-byte channel runs of 16 pixels.
// Results are bit-identical to conv3x3_halo_dma_kernel<2, true>'s where that kernel ran (same operands, same K order);
// STYLEX_S2D_DGRAD=0 selects the old kernels (A/B: tools/bench_s2d_dgrad.py; ablation builds: SD_PROBE_NO_STORE /
// SD_PROBE_NO_HALO / SD_STAGGER, DESIGN §3 "Round 5").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

// cache policy of the output stores (buffer_store aux bits; 2 = nt, streaming: tools/bench_s2d_dgrad.py A/B, DESIGN §3 "Round 5")
#ifndef SD_STORE_AUX
#define SD_STORE_AUX 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StylexS2dDgradArgs {
    int total_tiles;           // B * tiles_y * tiles_x * c_groups
    unsigned m_cg, m_tpi, m_tx;  // magic reciprocals of c_groups, tiles per image, tiles_x
};

namespace {

typedef StylexS2dDgradArgs SdArgs;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// block = NW waves: 8 (a 64-channel group in four sub-positions = 256 output channels, one block per CU) or 4 (a 32-channel
// group = 128 output channels, two blocks per CU); pixel tile 8 x 32 or 16 x 16 (TW = 16: images 16 pixels wide)
template <int TW_, int NW_>
struct SdCfg {
    static constexpr int TW = TW_, NW = NW_, TH = 256 / TW, HR = TH + 1, HC = TW + 1, NPX = HR * HC;  // halo: offsets 0, +1 only
    static constexpr int H_PIECES = (NPX + 15) / 16, H_STAGE = H_PIECES * 1024;  // DMA pieces of 16 pixel rows x 64 B
    static constexpr int CG = NW * 8;                                            // channels of a group (rows of a weight slot)
    static constexpr int W_ROWS = 9 * CG, W_PIECES = W_ROWS / 16, W_STAGE = W_PIECES * 1024;  // nine (sub-position, tap) slots
    static constexpr int W_BASE = 2 * H_STAGE, DUMP_BASE = W_BASE + 2 * W_STAGE, SMEM = DUMP_BASE + 1024;  // 112.6 / 75 KB
    static constexpr int HS = (H_PIECES + NW - 1) / NW, WS = (W_PIECES + NW - 1) / NW, NSLOT = HS + WS;  // DMA instructions per wave and stage
    static_assert(NSLOT == 8 || NSLOT == 10, "3 + 5 or 5 + 5 DMA instructions per wave and stage");
};
constexpr unsigned OOB = 0x80000000u;
#ifndef SD_EARLY_ISSUE
#define SD_EARLY_ISSUE 1
#endif
#ifdef SD_PROBE_NO_HALO
constexpr bool SD_NO_HALO = true;  // ablation build: the gradient halo is never fetched (zeros)
#else
constexpr bool SD_NO_HALO = false;
#endif
constexpr bool SD_EARLY = SD_EARLY_ISSUE != 0;  // 1: the whole DMA of the next stage goes out behind the first MFMAs of a stage

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* smem, int lds_off, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(smem + lds_off), 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int fastdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8& dst, int addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void mfma1(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// (sub-position, tap) slots of the staged weights, in LDS order: slot j holds rows [64 j, 64 j + 64)
//   j:    0     1     2     3     4     5     6     7     8
//   s:    0     1     1     2     2     3     3     3     3
//   tap:  4     4     5     4     7     4     5     7     8        (input offset dy = tap / 3 - 1, dx = tap % 3 - 1)
__host__ __device__ constexpr int slot_sub(int j) { return j == 0 ? 0 : j <= 2 ? 1 : j <= 4 ? 2 : 3; }
__host__ __device__ constexpr int slot_tap(int j) { return j == 2 || j == 6 ? 5 : j == 4 || j == 7 ? 7 : j == 8 ? 8 : 4; }

// Operand registers of a wave: the six halo fragments of one 16-channel k-step (3 halo rows x 2 column offsets),
// double-buffered by k-step; the weight fragments (this wave's 32-channel half) of the three slots of a step,
// double-buffered by step.  A step = three slots at one k-step = 6 MFMAs; a stage = 2 k-steps x 3 steps.
struct SdOps {
    bf16x8 b[2][3][2];
    bf16x8 a[2][3];
};

template <int KC>
__device__ __forceinline__ void read_b(SdOps& o, const int (&bb)[3][2], int lk) {
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int d = 0; d < 2; ++d)  // swizzle of row R = bb / 64 (stage strides are multiples of 1 KiB: (bb >> 8) & 3 is R's)
            lds_read16<0>(o.b[KC][h][d], bb[h][d] + (((KC * 2 + lk) ^ ((bb[h][d] >> 8) & 3)) << 4));
}
// slots 3 G .. 3 G + 2 at k-step KC into register set PAR
template <int CG, int G, int KC, int PAR>
__device__ __forceinline__ void read_a(SdOps& o, int ab, int af, int lk) {
    const int addr = ab + (((KC * 2 + lk) ^ af) << 4);
    lds_read16<((3 * G + 0) * CG) * 64>(o.a[PAR][0], addr);
    lds_read16<((3 * G + 1) * CG) * 64>(o.a[PAR][1], addr);
    lds_read16<((3 * G + 2) * CG) * 64>(o.a[PAR][2], addr);
}
__device__ __forceinline__ void wait_all(SdOps& o) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o.b[0][0][0]), "+v"(o.b[0][0][1]), "+v"(o.b[0][1][0]), "+v"(o.b[0][1][1]), "+v"(o.b[0][2][0]), "+v"(o.b[0][2][1]),
                   "+v"(o.b[1][0][0]), "+v"(o.b[1][0][1]), "+v"(o.b[1][1][0]), "+v"(o.b[1][1][1]), "+v"(o.b[1][2][0]), "+v"(o.b[1][2][1]),
                   "+v"(o.a[0][0]), "+v"(o.a[0][1]), "+v"(o.a[0][2]), "+v"(o.a[1][0]), "+v"(o.a[1][1]), "+v"(o.a[1][2]));
}
// the six MFMAs of slots 3 G .. 3 G + 2 at k-step KC: output rows 0, 1 of this wave's row pair
template <int G, int KC, int PAR>
__device__ __forceinline__ void step_mfma(f32x16 (&acc)[2][4], const SdOps& o) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
            const int j = 3 * G + q;
        const int sub = slot_sub(j), tap = slot_tap(j), dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) mfma1(acc[r][sub], o.a[PAR][q], o.b[KC][r + dy][dx]);  // D^T = W x X^T
    }
}

template <int TW, int NW>
__global__ __launch_bounds__(NW * 64, 2) void conv_s2d_dgrad_kernel(ConvKParams p, SdArgs sa) {
    using Cfg = SdCfg<TW, NW>;
    constexpr int TH = Cfg::TH, HC = Cfg::HC, NPX = Cfg::NPX, H_PIECES = Cfg::H_PIECES, H_STAGE = Cfg::H_STAGE, CG = Cfg::CG;
    constexpr int W_PIECES = Cfg::W_PIECES, W_STAGE = Cfg::W_STAGE, W_BASE = Cfg::W_BASE, DUMP_BASE = Cfg::DUMP_BASE;
    constexpr int HS = Cfg::HS, WS = Cfg::WS, NSLOT = Cfg::NSLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = __builtin_amdgcn_readfirstlane(p.Ho), W = __builtin_amdgcn_readfirstlane(p.Wo);
    const int NI = __builtin_amdgcn_readfirstlane(p.Ck);     // gz channels = K
    const int C = __builtin_amdgcn_readfirstlane(p.s2d_c);   // channels per sub-position; output pixel = 4 C channels
    const int cgs = C / CG, tiles_x = W / TW, tiles_y = H / TH, tpi = tiles_x * tiles_y;
    const int nch = NI >> 5;                                 // 32-channel K stages per tile

    // static tile list, XCD-contiguous (conv_pipe.hip): the channel groups of one pixel tile are neighbours
    const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int tq = sa.total_tiles >> 3, tr = sa.total_tiles & 7;
    const int xs = xcd * tq + (xcd < tr ? xcd : tr), xn = tq + (xcd < tr ? 1 : 0);
    if (bslot >= xn) return;
#ifdef SD_STAGGER
    // second residents of a CU (the upper half of the grid) start about half a tile late, so that the two blocks of a CU are
    // not in their store bursts at the same time
    if (NW == 4 && bslot >= (nslots >> 1))
        for (int i = 0; i < SD_STAGGER * (nch + 2); ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 clocks
#endif
    const int my_tiles = (xn - bslot + nslots - 1) / nslots;
    const int total = my_tiles * nch;
    auto decode = [&](int k, int& b, int& y0, int& x0, int& c0) {
        const int t = xs + bslot + k * nslots;
        int pt = fastdiv(t, sa.m_cg);
        c0 = (t - pt * cgs) * CG;
        b = fastdiv(pt, sa.m_tpi);
        pt -= b * tpi;
        const int ty = fastdiv(pt, sa.m_tx);
        y0 = ty * TH;
        x0 = (pt - ty * tiles_x) * TW;
    };

    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0x7ffffff0, 0x00020000);

    // ---- DMA lane constants: a piece = 16 rows x 64 B, lane -> row lane >> 2, physical slot lane & 3 = logical slot ^ swz(row)
    const int prow = lane >> 2;
    const unsigned lslot = (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
    unsigned vh[HS];  // halo: bits 1:0 = (bottom halo row, right halo column)
    int hdst[HS];
#pragma unroll
    for (int k = 0; k < HS; ++k) {
        const int piece = wave + NW * k;
        const int R = piece * 16 + prow;
        const int hr = R / HC, hc = R - hr * HC;
        const bool ok = piece < H_PIECES && R < NPX;
        vh[k] = ok ? (((unsigned)(hr * W + hc) * (unsigned)NI + lslot * 8u) * 2u) | (hr == TH ? 1u : 0u) | (hc == TW ? 2u : 0u) : OOB;
        hdst[k] = piece < H_PIECES ? piece * 1024 : -1;
    }
    unsigned vw[WS];
    int wdst[WS];
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        const int piece = wave + NW * k;
        const int wr = piece * 16 + prow;
        const int j = wr / CG, oc = wr % CG;
        const int s = j == 0 ? 0 : j <= 2 ? 1 : j <= 4 ? 2 : 3;
        const int tap = (j == 2 || j == 6) ? 5 : (j == 4 || j == 7) ? 7 : j == 8 ? 8 : 4;
        vw[k] = piece < W_PIECES ? (((unsigned)((s * C + oc) * 9 + (8 - tap)) * (unsigned)NI + lslot * 8u) * 2u) : OOB;
        wdst[k] = piece < W_PIECES ? piece * 1024 : -1;
    }

    // ---- producer cursor (uniform): stage (tile dk, chunk dch) the DMA stream fetches next
    int dk = 0, dch = 0, dslot = 0, db, dy0, dx0, dc0;
    decode(0, db, dy0, dx0, dc0);
#define SD_ISSUE(K)                                                                                                       \
    {                                                                                                                     \
        const bool more_ = dk < my_tiles;                                                                                 \
        if constexpr ((K) < HS) {                                                                                         \
            const unsigned soff_ = (unsigned)((((db * H + dy0) * W + dx0) * NI) + dch * 32) * 2u;                          \
            const unsigned edge_ = (dy0 + TH == H ? 1u : 0u) | (dx0 + TW == W ? 2u : 0u);                                 \
            const unsigned v_ = ((vh[(K) < HS ? (K) : 0] & edge_) || !more_ || SD_NO_HALO) ? OOB : (vh[(K) < HS ? (K) : 0] & ~15u); \
            dma16(rg, smem, hdst[(K) < HS ? (K) : 0] >= 0 ? dslot * H_STAGE + hdst[(K) < HS ? (K) : 0] : DUMP_BASE, v_,   \
                  more_ ? soff_ : 0u);                                                                                    \
        } else {                                                                                                          \
            constexpr int kw_ = (K) - HS < WS ? ((K) >= HS ? (K) - HS : 0) : 0;                                           \
            const unsigned soff_ = (unsigned)(dc0 * 9 * NI + dch * 32) * 2u;                                              \
            dma16(rw, smem, wdst[kw_] >= 0 ? W_BASE + dslot * W_STAGE + wdst[kw_] : DUMP_BASE, more_ ? vw[kw_] : OOB,      \
                  more_ ? soff_ : 0u);                                                                                    \
        }                                                                                                                 \
        if constexpr ((K) == NSLOT - 1) {                                                                                 \
            dslot ^= 1;                                                                                                   \
            if (++dch == nch) {                                                                                           \
                dch = 0;                                                                                                  \
                if (++dk < my_tiles) decode(dk, db, dy0, dx0, dc0);                                                       \
            }                                                                                                             \
        }                                                                                                                 \
    }

    // ---- wave roles and operand addressing: every wave runs the SAME instruction stream (a per-quartet tap split made the
    // accumulators a two-way phi and spilled 700 registers): row pair rp of the tile, 32-channel half jt of the 64-channel
    // group, all four sub-positions
    // (NW = 4: one 32-channel group per block, jt = 0).  A row group = two 32-pixel fragments: rows 2 rp + r of a 32-wide tile;
    // rows 4 rp + r + {0, 2} x 16 pixels of a 16-wide tile (interleaved so that fragment r one row down IS fragment r + 1)
    const int jt = wave >> 2, rp = wave & 3;
    const int li = lane & 31, lk = lane >> 5;
    const int prow0 = TW == 32 ? 2 * rp : 4 * rp + 2 * (li >> 4), pcol = TW == 32 ? li : (li & 15);
    int bb[3][2];  // halo fragment (row offset h, column offset d): byte offset of halo pixel (prow0 + h, pcol + d)
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int d = 0; d < 2; ++d) bb[h][d] = ((prow0 + h) * HC + pcol + d) * 64;
    int ab = W_BASE + (jt * 32 + li) * 64;  // weight row jt * 32 + li of a slot; (wr >> 2) & 3 == (li >> 2) & 3
    const int af = (li >> 2) & 3;

    f32x16 acc[2][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][s][e] = 0.f;
    };
    zero_acc();

    // ---- epilogue: output rows 2 rp, 2 rp + 1 of the tile, this wave's 32 channels in each of the four sub-positions.  A
    // lane holds, per (row, sub-position), the four 16-byte pieces (2 q + lk) of pixel li's 64-byte channel run; swapping
    // q with lane bit 4 (v_permlane16_swap) lets ONE store write the complete 64-byte runs of 16 pixels.
    const unsigned pixb = (unsigned)(4 * C) * 2u;  // bytes per output pixel
    const unsigned lane_off = (unsigned)(lane & 15) * pixb + (unsigned)(2 * ((lane >> 4) & 1) + (lane >> 5)) * 16u;
    const unsigned kk_off = TW == 32 ? 16u * pixb : 2u * (unsigned)W * pixb;  // second half of a fragment: pixels 16.. / two rows down
    auto pack2 = [](float a, float c) -> unsigned {
        f32x2_t t = {a, c};
        bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
        return *reinterpret_cast<unsigned*>(&r);
    };
    auto epilogue = [&](int b, int y0, int x0, int c0) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const unsigned soff = __builtin_amdgcn_readfirstlane(
                    (unsigned)((((b * H + y0 + (TW == 32 ? 2 : 4) * rp + r) * W + x0) * (4 * C)) + sub * C + c0 + jt * 32) * 2u);
                unsigned P[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    P[g][0] = pack2(acc[r][sub][4 * g + 0], acc[r][sub][4 * g + 1]);
                    P[g][1] = pack2(acc[r][sub][4 * g + 2], acc[r][sub][4 * g + 3]);
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        auto q = __builtin_amdgcn_permlane32_swap(P[g][h], P[g + 1][h], false, false);
                        P[g][h] = q[0];
                        P[g + 1][h] = q[1];
                    }
                u32x4 R[2];  // R[q]: channels 16 q + 8 lk .. + 7 of pixel li
#pragma unroll
                for (int q = 0; q < 2; ++q) R[q] = u32x4{P[2 * q][0], P[2 * q][1], P[2 * q + 1][0], P[2 * q + 1][1]};
#pragma unroll
                for (int d = 0; d < 4; ++d) {  // q <-> lane bit 4: R[0] = pixels 0..15, R[1] = pixels 16..31, lane = (lk, q, pixel % 16)
                    auto t = __builtin_amdgcn_permlane16_swap(R[0][d], R[1][d], false, false);
                    R[0][d] = t[0];
                    R[1][d] = t[1];
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#ifndef SD_PROBE_NO_STORE  // (ablation builds: tools/bench_s2d_dgrad.py with STYLEX_HIP_LIB)
                    __builtin_amdgcn_raw_buffer_store_b128(R[kk], ry, lane_off, soff + (unsigned)kk * kk_off, SD_STORE_AUX);
#endif
                    asm volatile("s_nop 1" : "+v"(R[kk]) : : "memory");  // VMEM store data hazard (conv_pipe.hip)
                }
            }
    };

    // ---- prologue: stage 0
    SD_ISSUE(0) SD_ISSUE(1) SD_ISSUE(2) SD_ISSUE(3) SD_ISSUE(4) SD_ISSUE(5) SD_ISSUE(6) SD_ISSUE(7)
    if constexpr (NSLOT > 8) { SD_ISSUE(8 < NSLOT ? 8 : 0) SD_ISSUE(9 < NSLOT ? 9 : 0) }
    SdOps o{};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cslot = 0, c_ch = 0, c_k = 0, cb, cy0, cx0, cc0;
    decode(0, cb, cy0, cx0, cc0);
    // one step = three slots at one k-step: wait for its operands (requested a step ago), request the next step's weight
    // fragments (and, at the first step, the halo fragments of k-step 1), six MFMAs, two DMA pieces
#define SD_STEP(KC_, G_)                                                                                 \
    {                                                                                                    \
        constexpr int S_ = (KC_) * 3 + (G_), GN_ = ((G_) + 1) % 3, KN_ = (G_) == 2 ? 1 : (KC_);          \
        wait_all(o);                                                                                     \
        if constexpr (S_ == 0) read_b<1>(o, bb, lk);                                                     \
        read_a<CG, GN_, KN_, (S_ + 1) & 1>(o, ab, af, lk);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        step_mfma<G_, KC_, S_ & 1>(acc, o);                                                              \
        if constexpr (SD_EARLY) {                                                                        \
            if constexpr (S_ == 0) if (!pre_issued) {                                                    \
                SD_ISSUE(0) SD_ISSUE(1) SD_ISSUE(2) SD_ISSUE(3) SD_ISSUE(4) SD_ISSUE(5) SD_ISSUE(6) SD_ISSUE(7) \
                if constexpr (NSLOT > 8) { SD_ISSUE(8 < NSLOT ? 8 : 0) SD_ISSUE(9 < NSLOT ? 9 : 0) }     \
            }                                                                                            \
        } else {                                                                                         \
            if constexpr (2 * S_ < NSLOT) { SD_ISSUE(2 * S_ < NSLOT ? 2 * S_ : 0) }                      \
            if constexpr (2 * S_ + 1 < NSLOT) { SD_ISSUE(2 * S_ + 1 < NSLOT ? 2 * S_ + 1 : 0) }          \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    }
    // the last step of a stage (k-step 1, slots 6..8): the barrier that publishes the next stage sits in front of its MFMAs
#define SD_LAST()                                                                                        \
    {                                                                                                    \
        wait_all(o);                                                                                     \
        if (pre_issued) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); /* the 16 stores of the last tile stay in flight */ \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
        pre_issued = false;                                                                              \
        __builtin_amdgcn_s_barrier();                                                                    \
        asm volatile("" ::: "memory");                                                                   \
        {                                                                                                \
            const int dh_ = cslot ? -H_STAGE : H_STAGE, dw_ = cslot ? -W_STAGE : W_STAGE;                \
            cslot ^= 1;                                                                                  \
            _Pragma("unroll") for (int h_ = 0; h_ < 3; ++h_) { bb[h_][0] += dh_; bb[h_][1] += dh_; }     \
            ab += dw_;                                                                                   \
        }                                                                                                \
        read_b<0>(o, bb, lk);                                                                            \
        read_a<CG, 0, 0, 0>(o, ab, af, lk);                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        step_mfma<2, 1, 1>(acc, o);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    }

    // Stores and the stage barrier.  Vector-memory operations complete in issue order (vmcnt), so a stage barrier that waits
    // for its DMA also waits for every store issued before it: with the stores of a tile issued between two stages' DMA, the
    // first barrier of the next tile waited for the write path (ablation without the stores: 64 -> 64 @256^2 0.21 -> 0.09 ms,
    // more than the 0.08 ms the bytes take at the fill rate).  So the DMA of the stage AFTER the next goes out at the end of
    // a tile, ahead of the 16 stores (its ring slot is free once the tile's last barrier has passed), the next stage issues
    // nothing and its barrier waits with vmcnt(16): everything but the stores.
    static_assert(SD_EARLY, "the tile-end pre-issue replaces the step-0 issue of the early-issue schedule");
    bool pre_issued = false;
    // first operands of stage 0
    read_b<0>(o, bb, lk);
    read_a<CG, 0, 0, 0>(o, ab, af, lk);
    for (int g = 0; g < total; ++g) {
        SD_STEP(0, 0) SD_STEP(0, 1) SD_STEP(0, 2) SD_STEP(1, 0) SD_STEP(1, 1)
        SD_LAST()
        if (++c_ch == nch) {  // tile complete
            c_ch = 0;
            {  // the DMA of the stage after the next one, ahead of this tile's stores
                SD_ISSUE(0) SD_ISSUE(1) SD_ISSUE(2) SD_ISSUE(3) SD_ISSUE(4) SD_ISSUE(5) SD_ISSUE(6) SD_ISSUE(7)
                if constexpr (NSLOT > 8) { SD_ISSUE(8 < NSLOT ? 8 : 0) SD_ISSUE(9 < NSLOT ? 9 : 0) }
                pre_issued = true;
            }
            asm volatile("s_nop 15\n\ts_nop 15"
                         : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                           "+v"(acc[1][2]), "+v"(acc[1][3]));
            epilogue(cb, cy0, cx0, cc0);
            if (++c_k < my_tiles) {
                decode(c_k, cb, cy0, cx0, cc0);
                zero_acc();
            }
        }
    }
#undef SD_LAST
#undef SD_STEP
#undef SD_ISSUE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs must not outlive the block's LDS allocation
}

int g_sd_cus = 0;
unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

template <int TW, int NW>
int launch_sd(const ConvKParams& p, hipStream_t s) {
    using Cfg = SdCfg<TW, NW>;
    static int attr_state = 0;
    if (attr_state == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_dgrad_kernel<TW, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr_state = e == hipSuccess ? 1 : -1;
    }
    if (attr_state < 0) return STYLEX_NOT_APPLICABLE;
    if (!g_sd_cus) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        g_sd_cus = n > 0 ? (n & ~7) : 256;
        if (g_sd_cus < 8) g_sd_cus = 8;
    }
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / Cfg::TH, cgs = p.s2d_c / Cfg::CG;
    SdArgs sa;
    sa.total_tiles = p.B * tiles_x * tiles_y * cgs;
    sa.m_cg = magic_of(cgs);
    sa.m_tpi = magic_of(tiles_x * tiles_y);
    sa.m_tx = magic_of(tiles_x);
    stylex_note_kernel("conv_s2d_dgrad_kernel<%d, %d>", TW, NW);
    const int blocks = NW == 4 ? 2 * g_sd_cus : g_sd_cus;  // 4-wave blocks: two per CU
    hipLaunchKernelGGL((conv_s2d_dgrad_kernel<TW, NW>), dim3((unsigned)blocks), dim3(NW * 64), Cfg::SMEM, s, p, sa);
    return (int)hipGetLastError();
}

}  // namespace

// data gradient of the space-to-depth stride-2 conv: bf16, no epilogue, whole 8 x 32 or 16 x 16 tiles, 32- / 64-channel groups,
// 32-channel K stages
int stylex_launch_s2d_dgrad(const ConvKParams& p, hipStream_t s) {
    const char* env = getenv("STYLEX_S2D_DGRAD");  // read per launch: A/B tests toggle it in-process
    if (env && env[0] == '0') return STYLEX_NOT_APPLICABLE;
    if (!p.act_bf16 || !p.s2d_c || !p.flip_taps || p.flags || p.a_scale || p.x2 || p.mask || p.gate_mask) return STYLEX_NOT_APPLICABLE;
    if (p.N != 4 * p.s2d_c || p.s2d_c % 32 != 0 || p.Ck % 32 != 0 || p.Ck < 64) return STYLEX_NOT_APPLICABLE;
    const bool w32 = p.Wo % 32 == 0 && p.Ho % 8 == 0, w16 = p.Wo % 16 == 0 && p.Ho % 16 == 0;
    if (!w32 && !w16) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (reinterpret_cast<uintptr_t>(p.y) & 15))
        return STYLEX_NOT_APPLICABLE;
    if ((long)p.B * p.Ho * p.Wo * p.Ck * 2 >= (1l << 30) || (long)p.B * p.Ho * p.Wo * p.N * 2 >= (1l << 31) - 16 ||
        (long)p.N * 9 * p.Ck * 2 >= (1l << 30))
        return STYLEX_NOT_APPLICABLE;
    if (p.dry) return 0;
    // STYLEX_S2D_DGRAD_TILE=1: 8-wave blocks (64-channel groups, one block per CU) instead of the 4-wave default
    const char* te = getenv("STYLEX_S2D_DGRAD_TILE");
    if (te && te[0] == '1' && p.s2d_c % 64 == 0) return w32 ? launch_sd<32, 8>(p, s) : launch_sd<16, 8>(p, s);
    return w32 ? launch_sd<32, 4>(p, s) : launch_sd<16, 4>(p, s);
}
