// conv_s2d_fwd.hip — round 5: forward of the 3x3 / stride-2 conv of a DiscriminatorBlock
// (/root/reference/stylex/stylex_train.py:733-736, merge :741-743) in its space-to-depth form, as ONE pipelined K loop.
//
//   y[b][oy][ox][n] = sum_{s} sum_{t in F(s)} sum_c x2[b][oy + dy(t)][ox + dx(t)][s*C + c] * Wf[n][t][s*C + c]
//                     (+ sum_c2 xr[b][oy][ox][c2] * w2[n][c2])            -- the block's 1x1 / stride-2 residual conv
//   out = (y + bias (+ residual)) * res_scale
//
// (x2 = the blurred block input stored space-to-depth, [B][H][W][4C]; F(s) = the 1, 2, 2, 4 taps of input sub-position
// s = (sy, sx), offsets dy, dx in {-1, 0}; Wf = stylex_pack_weight_s2d's forward pack [N][9][4C].)
//
// Why a kernel of its own.  conv3x3_halo_dma_kernel<2, true> runs the four sub-positions as four chunk loops of 16 channels
// with a vmcnt(0) + barrier drain per 8-32 MFMAs per wave and a 64-channel output tile per block (two blocks per CU);
// 16-pixel-wide images (512 -> 512 @32^2 -> 16^2) fall to the register-staged kernel at 0.26-0.45 PF.  Here: persistent
// blocks walk a static tile list; a tile = 256 pixels x 128 channels in a 4-wave block, two blocks per CU (default), the
// same tile on 8 waves with a three-deep ring for launches of at most one tile per CU, 256 x 64 for N = 64, or 256 x 256 /
// 512 x 128 in an 8-wave block (STYLEX_S2D_FWD_TILE); 8 (4) accumulator tiles per wave.  The K loop is a sequence of PHASES
// — [residual 1x1 segment,] s0 {(0,0)}, s1 {(0,-1), (0,0)}, s2 {(-1,0), (0,0)},
// s3 {(-1,-1), (-1,0)}, s3 {(0,-1), (0,0)} — each a run of 32-channel stages: the phase's input halo (offsets -1 / 0 only:
// top row + left column) and its one or two tap slots of weights stream through two-deep LDS-DMA rings that are never
// drained, across phase and tile boundaries; the next stage's DMA is issued behind the first MFMAs of a stage, one barrier
// per stage in front of its last MFMAs (conv_s2d_dgrad.hip / conv_wgrad_pipe.hip's recipe).  Every phase has its own
// instantiation of the stage loop (tap offsets are compile-time), run one after the other per tile; the DMA producer is
// phase-agnostic scalar bookkeeping.  64-byte LDS rows, 16-byte slot q of row R at q ^ ((R >> 2) & 3).  Epilogue from
// registers (bias, merge, v_permlane32_swap + v_permlane16_swap, 64-byte channel runs of 16 pixels per store).
// STYLEX_S2D_FWD=0 selects the old kernels (A/B: tools/bench_s2d_fwd.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "stylex_internal.h"

// cache policy of the output stores (buffer_store aux bits; 2 = nt, streaming: tools/bench_s2d_dgrad.py A/B, DESIGN §3 "Round 5")
#ifndef SF_STORE_AUX
#define SF_STORE_AUX 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StylexS2dFwdArgs {
    int total_tiles;             // B * tiles_y * tiles_x * n_groups
    unsigned m_ng, m_tpi, m_tx;  // magic reciprocals of n_groups, tiles per image, tiles_x
};

namespace {

typedef StylexS2dFwdArgs SfArgs;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
constexpr unsigned OOB = 0x80000000u;

// block tile: NW waves x 2 NF accumulator tiles = PXF 32-pixel fragments x NT output channels; TW = tile width in pixels.
// <*, 128, 4, 4>: 256 px x 128 n in a 4-wave block, two blocks per CU (default); <*, 64, 4, 2>: 256 px x 64 n likewise (N = 64);
// <*, 128, 8, 2>: the 256 x 128 tile on 8 waves
// (launches of at most one tile per CU); <*, 256, 8, 4>: 256 px x 256 n; <32, 128, 8, 4>: 512 px x 128 n
template <int TW_, int NT_, int NW_, int NF_>
struct FwCfg {
    static constexpr int TW = TW_, NT = NT_, NW = NW_, NF = NF_;
    // ring depth: the 8-wave / 2-fragment variant runs alone on its CU (launches of at most one tile per CU), where a stage is
    // as long as a DMA round trip: its ring is three deep (the DMA of stage g + 2 is issued in stage g, counted vmcnt)
    static constexpr int RD = (NW_ == 8 && NF_ == 2) ? 3 : 2;
    static constexpr int NG = NT / (32 * NF), RPN = NW / NG;  // channel groups of a block, row groups (waves per channel group)
    static constexpr int PXF = RPN * 2;
    static constexpr int TH = PXF * 32 / TW;
    static constexpr int HR = TH + 1, HC = TW + 1, NPX = HR * HC;  // halo: one row above, one column to the left
    static constexpr int H_PIECES = (NPX + 15) / 16, H_STAGE = H_PIECES * 1024;  // DMA pieces of 16 pixel rows x 64 B
    static constexpr int W_PIECES = 2 * NT / 16, W_STAGE = W_PIECES * 1024;     // two tap slots x NT rows
    static constexpr int W_BASE = RD * H_STAGE, DUMP_BASE = W_BASE + RD * W_STAGE, SMEM = DUMP_BASE + 1024;
    static constexpr int HS = (H_PIECES + NW - 1) / NW, WS = W_PIECES / NW, NDMA = HS + WS;  // DMA instructions per wave and stage
    static_assert(NDMA == 5 || NDMA == 7 || NDMA == 9, "3 + 2, 3 + 4, 5 + 2 or 5 + 4 DMA instructions per wave and stage");
    static_assert(W_PIECES % (2 * NW) == 0, "a DMA instruction stays inside one tap slot");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

// a phase of the K loop: NS tap slots with input offsets (DY, DX) in {-1, 0}
template <int NS_, int DY0_, int DX0_, int DY1_, int DX1_>
struct Phase {
    static constexpr int NS = NS_;
    static constexpr int dy(int j) { return j == 0 ? DY0_ : DY1_; }
    static constexpr int dx(int j) { return j == 0 ? DX0_ : DX1_; }
    // halo fragments a wave needs: rows h = r + dy + 1 (r = 0, 1: the wave's two pixel fragments), column offsets d = dx + 1
    static constexpr int hmin = ((NS_ == 2 && DY1_ < DY0_) ? DY1_ : DY0_) + 1;
    static constexpr int hmax = ((NS_ == 2 && DY1_ > DY0_) ? DY1_ : DY0_) + 2;
    static constexpr int dmin = ((NS_ == 2 && DX1_ < DX0_) ? DX1_ : DX0_) + 1;
    static constexpr int dmax = ((NS_ == 2 && DX1_ > DX0_) ? DX1_ : DX0_) + 1;
    static constexpr int nd = dmax - dmin + 1, nh = hmax - hmin + 1, NB = nh * nd;
    static constexpr int frag(int h, int d) { return (h - hmin) * nd + (d - dmin); }
    static_assert(NB <= 4, "at most four halo fragments per k-step");
};
typedef Phase<1, 0, 0, 0, 0> PhCentre;     // residual segment, sub-position 0
typedef Phase<2, 0, -1, 0, 0> PhLeft;      // sub-position 1 (taps 3, 4) and the second half of sub-position 3
typedef Phase<2, -1, 0, 0, 0> PhUp;        // sub-position 2 (taps 1, 4)
typedef Phase<2, -1, -1, -1, 0> PhUpLeft;  // first half of sub-position 3 (taps 0, 1)

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

// Operand registers of a wave: up to four halo fragments per 16-channel k-step, double-buffered by k-step; the four weight
// fragments (this wave's four 32-channel output groups) of one (slot, k-step) step, double-buffered by step.
struct SfOps {
    bf16x8 b[2][4];
    bf16x8 a[2][4];
};

template <class P, int KC>
__device__ __forceinline__ void read_b(SfOps& o, const int (&bb)[3][2], int lk) {
#pragma unroll
    for (int i = 0; i < P::NB; ++i) {
        const int h = P::hmin + i / P::nd, d = P::dmin + i % P::nd;
        // swizzle of row R = bb / 64 (stage strides are multiples of 1 KiB: (bb >> 8) & 3 is R's)
        lds_read16<0>(o.b[KC][i], bb[h][d] + (((KC * 2 + lk) ^ ((bb[h][d] >> 8) & 3)) << 4));
    }
}
template <int NT, int NF, int J, int KC, int PAR>
__device__ __forceinline__ void read_a(SfOps& o, int ab, int af, int lk) {
    const int addr = ab + (((KC * 2 + lk) ^ af) << 4);
    lds_read16<(J * NT + 0) * 64>(o.a[PAR][0], addr);
    lds_read16<(J * NT + 32) * 64>(o.a[PAR][1], addr);
    if constexpr (NF == 4) {
        lds_read16<(J * NT + 64) * 64>(o.a[PAR][2], addr);
        lds_read16<(J * NT + 96) * 64>(o.a[PAR][3], addr);
    }
}
__device__ __forceinline__ void wait_all(SfOps& o) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o.b[0][0]), "+v"(o.b[0][1]), "+v"(o.b[0][2]), "+v"(o.b[0][3]), "+v"(o.b[1][0]), "+v"(o.b[1][1]),
                   "+v"(o.b[1][2]), "+v"(o.b[1][3]), "+v"(o.a[0][0]), "+v"(o.a[0][1]), "+v"(o.a[0][2]), "+v"(o.a[0][3]),
                   "+v"(o.a[1][0]), "+v"(o.a[1][1]), "+v"(o.a[1][2]), "+v"(o.a[1][3]));
}
// the 2 NF MFMAs of tap slot J at k-step KC: two pixel fragments x NF 32-channel groups
template <class P, int NF, int J, int KC, int PAR>
__device__ __forceinline__ void step_mfma(f32x16 (&acc)[2][NF], const SfOps& o) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 2; ++r) mfma1(acc[r][nf], o.a[PAR][nf], o.b[KC][P::frag(r + P::dy(J) + 1, P::dx(J) + 1)]);  // D^T = W x X^T
}

template <int TW, int NT, int NW, int NF>
__global__ __launch_bounds__(NW * 64, 2) void conv_s2d_fwd_kernel(ConvKParams p, SfArgs sa) {
    using Cfg = FwCfg<TW, NT, NW, NF>;
    constexpr int HC = Cfg::HC, H_STAGE = Cfg::H_STAGE, W_STAGE = Cfg::W_STAGE, W_BASE = Cfg::W_BASE, DUMP_BASE = Cfg::DUMP_BASE;
    constexpr int HS = Cfg::HS, WS = Cfg::WS, NDMA = Cfg::NDMA, TH = Cfg::TH, RD = Cfg::RD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = __builtin_amdgcn_readfirstlane(p.Ho), W = __builtin_amdgcn_readfirstlane(p.Wo);
    const int N = __builtin_amdgcn_readfirstlane(p.N);
    const int C = __builtin_amdgcn_readfirstlane(p.s2d_c);  // channels per input sub-position; an input pixel = 4 C channels
    const int c2 = p.x2 ? __builtin_amdgcn_readfirstlane(p.c2) : 0;
    const int CK = 4 * C;
    const int ngs = N / NT, tiles_x = W / TW, tiles_y = H / TH, tpi = tiles_x * tiles_y;
    const int nchC = C >> 5, nch2 = (c2 + 31) >> 5;  // 32-channel stages per main phase / of the residual phase (last one ragged)
    const int first_ph = nch2 ? 0 : 1;
    const int stages_per_tile = nch2 + 5 * nchC;

    // static tile list, XCD-contiguous (conv_pipe.hip): the channel groups of one pixel tile are neighbours
    const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int tq = sa.total_tiles >> 3, tr = sa.total_tiles & 7;
    const int xs = xcd * tq + (xcd < tr ? xcd : tr), xn = tq + (xcd < tr ? 1 : 0);
    if (bslot >= xn) return;
    const int my_tiles = (xn - bslot + nslots - 1) / nslots;
    auto decode = [&](int k, int& b, int& y0, int& x0, int& n0) {
        const int t = xs + bslot + k * nslots;
        int pt = fastdiv(t, sa.m_ng);
        n0 = (t - pt * ngs) * NT;
        b = fastdiv(pt, sa.m_tpi);
        pt -= b * tpi;
        const int ty = fastdiv(pt, sa.m_tx);
        y0 = ty * TH;
        x0 = (pt - ty * tiles_x) * TW;
    };

    // input descriptors are based one row + one pixel BEFORE the tensor, so that the halo origin (y0 - 1, x0 - 1) of every
    // tile is a non-negative offset; out-of-image halo pixels are never fetched (OOB voffset -> zeros)
    const char* xa = reinterpret_cast<const char*>(p.a) - (long)(W + 1) * CK * 2;
    const char* xr = p.x2 ? reinterpret_cast<const char*>(p.x2) - (long)(W + 1) * c2 * 2 : xa;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xa), 0, 0x7ffffff0, 0x00020000);  // < 2 GiB: OOB stays out of range
    const __amdgpu_buffer_rsrc_t rxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xr), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x2 ? p.w2 : p.w), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0x7ffffff0, 0x00020000);

    // ---- DMA lane constants: a piece = 16 rows x 64 B, lane -> row lane >> 2, physical slot lane & 3 = logical slot ^ swz(row)
    const int prow = lane >> 2;
    const unsigned lslot = (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
    int hpix[HS];  // halo pixel hr * W + hc of this lane's row | bit 30: top halo row | bit 29: left halo column; < 0: none
    int hdst[HS];
#pragma unroll
    for (int k = 0; k < HS; ++k) {
        const int piece = wave + NW * k;
        const int R = piece * 16 + prow;
        const int hr = R / HC, hc = R - hr * HC;
        const bool ok = piece < Cfg::H_PIECES && R < Cfg::NPX;
        hpix[k] = ok ? ((hr * W + hc) | (hr == 0 ? (1 << 30) : 0) | (hc == 0 ? (1 << 29) : 0)) : -1;
        hdst[k] = piece < Cfg::H_PIECES ? piece * 1024 : -1;
    }
    int wrow[WS];  // output channel (row of the weight tile) this lane fetches in DMA instruction k; its tap slot is k / (WS / 2)
#pragma unroll
    for (int k = 0; k < WS; ++k) wrow[k] = ((wave + NW * k) * 16 + prow) & (NT - 1);

    // ---- producer cursor (uniform): the stage (tile dk, phase dph, chunk dch) the DMA stream fetches next
    int dk = 0, dph = first_ph, dch = 0, dslot = 0, db, dy0, dx0, dn0;
    decode(0, db, dy0, dx0, dn0);
    auto issue = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        const bool more = dk < my_tiles;
        if constexpr (K < HS) {
            const int ps = dph == 0 ? c2 * 2 : CK * 2;  // bytes per input pixel
            const int sub = dph == 0 ? 0 : (dph >= 4 ? 3 : dph - 1);
            const unsigned soff = (unsigned)(((db * H + dy0) * W + dx0) * ps + (dph == 0 ? 0 : sub * C * 2) + dch * 64);
            const int edge = (dy0 == 0 ? (1 << 30) : 0) | (dx0 == 0 ? (1 << 29) : 0);
            const int hp = hpix[K];
            // (residual segment: channels past c2 in its last 32-channel stage are not fetched)
            const bool cok = dph != 0 || (int)(lslot * 8u) + dch * 32 < c2;
            const unsigned v = (hp < 0 || (hp & edge) || !more || !cok) ? OOB : (unsigned)((hp & 0x1fffffff) * ps) + lslot * 16u;
            const int dst = hdst[K] >= 0 ? dslot * H_STAGE + hdst[K] : DUMP_BASE;
            if (dph == 0) dma16(rxr, smem, dst, v, more ? soff : 0u);
            else dma16(rx, smem, dst, v, more ? soff : 0u);
        } else {
            constexpr int kw = K - HS, slot = kw / (WS / 2);
            // tap of (phase, slot): s0 {4}; s1 {3, 4}; s2 {1, 4}; s3 {0, 1} then {3, 4}
            const int tap = slot == 1 ? (dph == 4 ? 1 : 4) : (dph == 2 || dph == 5 ? 3 : dph == 3 ? 1 : dph == 4 ? 0 : 4);
            const int sub = dph >= 4 ? 3 : dph - 1;
            const bool live = more && (slot == 0 || dph >= 2);
            const int dst = W_BASE + dslot * W_STAGE + (wave + NW * kw) * 1024;
            if (dph == 0) {
                const unsigned v = (live && (int)(lslot * 8u) + dch * 32 < c2) ? (unsigned)(wrow[kw] * c2 * 2) + lslot * 16u : OOB;
                dma16(rw2, smem, dst, v, live ? (unsigned)(dn0 * c2 * 2 + dch * 64) : 0u);
            } else {
                const unsigned v = live ? (unsigned)(wrow[kw] * 9 * CK * 2) + lslot * 16u : OOB;
                dma16(rw, smem, dst, v, live ? (unsigned)((dn0 * 9 + tap) * CK * 2 + sub * C * 2 + dch * 64) : 0u);
            }
        }
        if constexpr (K == NDMA - 1) {
            dslot = dslot + 1 == RD ? 0 : dslot + 1;
            if (++dch == (dph == 0 ? nch2 : nchC)) {
                dch = 0;
                if (++dph == 6) {
                    dph = first_ph;
                    if (++dk < my_tiles) decode(dk, db, dy0, dx0, dn0);
                }
            }
        }
    };
    // DMA instructions [K0, K1) of the next stage
    auto issue_range = [&](auto k0, auto k1) {
        constexpr int K0 = decltype(k0)::value, K1 = decltype(k1)::value;
        if constexpr (K0 + 0 < K1) issue(std::integral_constant<int, K0 + 0>{});
        if constexpr (K0 + 1 < K1) issue(std::integral_constant<int, K0 + 1>{});
        if constexpr (K0 + 2 < K1) issue(std::integral_constant<int, K0 + 2>{});
        if constexpr (K0 + 3 < K1) issue(std::integral_constant<int, K0 + 3>{});
        if constexpr (K0 + 4 < K1) issue(std::integral_constant<int, K0 + 4>{});
        if constexpr (K0 + 5 < K1) issue(std::integral_constant<int, K0 + 5>{});
        if constexpr (K0 + 6 < K1) issue(std::integral_constant<int, K0 + 6>{});
        if constexpr (K0 + 7 < K1) issue(std::integral_constant<int, K0 + 7>{});
        if constexpr (K0 + 8 < K1) issue(std::integral_constant<int, K0 + 8>{});
    };
    using D0 = std::integral_constant<int, 0>;
#ifndef SF_EARLY_ISSUE
#define SF_EARLY_ISSUE 1
#endif
    // a two-slot stage issues [0, D1), [D1, D2), [D2, NDMA) behind its first three steps (SF_EARLY_ISSUE: all behind the first)
    using D1 = std::integral_constant<int, SF_EARLY_ISSUE ? NDMA : 3>;
    using D2 = std::integral_constant<int, SF_EARLY_ISSUE ? NDMA : (NDMA == 7 ? 5 : 6)>;
    using DN = std::integral_constant<int, NDMA>;

    // ---- wave roles and operand addressing: every wave runs the same instruction stream.  NT = 256: row group rp = wave & 3,
    // 128-channel half jt = wave >> 2; NT = 128: row group rp = wave, all 128 channels.  A row group = two 32-pixel
    // fragments (NW = 4: rp = wave, all 128 channels): rows 2 rp + r of a 32-wide tile; rows 4 rp + r + {0, 2} x 16 pixels of a 16-wide tile (interleaved so that
    // fragment r shifted down by one row IS fragment r + 1: three halo rows serve both fragments at offsets -1 / 0).
    const int jt = wave / Cfg::RPN, rp = wave % Cfg::RPN;  // channel group (32 NF channels) and row group of this wave
    const int li = lane & 31, lk = lane >> 5;
    const int prow0 = TW == 32 ? 2 * rp : 4 * rp + 2 * (li >> 4), pcol = TW == 32 ? li : (li & 15);
    int bb[3][2];  // halo fragment (row offset h, column offset d): byte offset of halo pixel (prow0 + h, pcol + d)
#pragma unroll
    for (int h = 0; h < 3; ++h)
#pragma unroll
        for (int d = 0; d < 2; ++d) bb[h][d] = ((prow0 + h) * HC + pcol + d) * 64;
    int ab = W_BASE + (jt * 32 * NF + li) * 64;  // weight row jt * 32 NF + li of slot 0; ((row >> 2) & 3) == (li >> 2) & 3
    const int af = (li >> 2) & 3;

    f32x16 acc[2][NF];
    auto zero_acc = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][nf][e] = 0.f;
    };
    zero_acc();

    // ---- epilogue: (acc + bias (+ residual)) * res_scale, pixel fragment r x 32-channel group nf at a time.  A lane holds the
    // four 16-byte pieces (2 q + lk) of pixel li's 64-byte channel run; swapping q with lane bit 4 (v_permlane16_swap) lets one
    // store write the complete runs of 16 pixels.
    const unsigned pixb = (unsigned)N * 2u;  // bytes per output pixel
    const unsigned lane_off = (unsigned)(lane & 15) * pixb + (unsigned)(2 * ((lane >> 4) & 1) + (lane >> 5)) * 16u;
    const unsigned kk_off = TW == 32 ? 16u * pixb : 2u * (unsigned)W * pixb;  // second half of a fragment: pixels 16.. / two rows down
    const bool has_bias = (p.flags & STYLEX_EPI_BIAS) != 0;
    const unsigned short* res = (p.flags & STYLEX_EPI_RESIDUAL) ? reinterpret_cast<const unsigned short*>(p.residual) : nullptr;
    const float rsc = (res || p.x2) ? p.res_scale : 1.f;
    auto pack2 = [](float a, float c) -> unsigned {
        f32x2_t t = {a, c};
        bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
        return *reinterpret_cast<unsigned*>(&r);
    };
    auto epilogue = [&](int b, int y0, int x0, int n0) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int nb = n0 + jt * 32 * NF + nf * 32;  // uniform
            float4 b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b4[g] = has_bias ? *reinterpret_cast<const float4*>(p.bias + nb + 8 * g + 4 * lk) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row0 = y0 + (TW == 32 ? 2 * rp : 4 * rp) + r;  // uniform: the fragment's first row
                const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(((b * H + row0) * W + x0) * N + nb) * 2u);
                unsigned P[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[r][nf][4 * g + 0] + b4[g].x, v1 = acc[r][nf][4 * g + 1] + b4[g].y;
                    float v2 = acc[r][nf][4 * g + 2] + b4[g].z, v3 = acc[r][nf][4 * g + 3] + b4[g].w;
                    if (res) {  // this lane's pixel before the swaps: li of the fragment
                        const int py = TW == 32 ? row0 : row0 + 2 * (li >> 4), px = x0 + pcol;
                        const uint2 rv = *reinterpret_cast<const uint2*>(res + ((long)(b * H + py) * W + px) * N + nb + 8 * g + 4 * lk);
                        v0 += __uint_as_float(rv.x << 16);
                        v1 += __uint_as_float(rv.x & 0xffff0000u);
                        v2 += __uint_as_float(rv.y << 16);
                        v3 += __uint_as_float(rv.y & 0xffff0000u);
                    }
                    P[g][0] = pack2(v0 * rsc, v1 * rsc);
                    P[g][1] = pack2(v2 * rsc, v3 * rsc);
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
                    __builtin_amdgcn_raw_buffer_store_b128(R[kk], ry, lane_off, soff + (unsigned)kk * kk_off, SF_STORE_AUX);
                    asm volatile("s_nop 1" : "+v"(R[kk]) : : "memory");  // VMEM store data hazard (conv_pipe.hip)
                }
            }
        }
    };

    // ---- prologue: stage 0
    issue_range(D0{}, DN{});
    if constexpr (RD == 3) issue_range(D0{}, DN{});  // stage 1
    SfOps o{};
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cslot = 0;
    // the barrier that publishes the next stage (in front of a stage's last MFMAs), and the ring step of the consumer
    auto publish = [&]() {
        // everything but the newest stage of a three-deep ring (vector-memory operations complete in issue order)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool wrap = cslot == RD - 1;
        const int dh = wrap ? -(RD - 1) * H_STAGE : H_STAGE, dw = wrap ? -(RD - 1) * W_STAGE : W_STAGE;
        cslot = wrap ? 0 : cslot + 1;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            bb[h][0] += dh;
            bb[h][1] += dh;
        }
        ab += dw;
    };
    // one phase: n stages of its instantiation of the stage loop.  Steps of a two-slot stage: (k-step 0, slot 0), (0, 1),
    // (1, 0), (1, 1); of a one-slot stage: (0, 0), (1, 0).  A step waits for its operands (requested a step ago), requests
    // the next step's, runs its eight MFMAs and issues its share of the next stage's DMA.
    auto run = [&](auto ph, int n) {
        using P = decltype(ph);
        read_b<P, 0>(o, bb, lk);
        read_a<NT, NF, 0, 0, 0>(o, ab, af, lk);
        for (int g = 0; g < n; ++g) {
            if constexpr (P::NS == 2) {
                wait_all(o);
                read_b<P, 1>(o, bb, lk);
                read_a<NT, NF, 1, 0, 1>(o, ab, af, lk);
                __builtin_amdgcn_sched_barrier(0);
                step_mfma<P, NF, 0, 0, 0>(acc, o);
                issue_range(D0{}, D1{});
                __builtin_amdgcn_sched_barrier(0);
                wait_all(o);
                read_a<NT, NF, 0, 1, 0>(o, ab, af, lk);
                __builtin_amdgcn_sched_barrier(0);
                step_mfma<P, NF, 1, 0, 1>(acc, o);
                issue_range(D1{}, D2{});
                __builtin_amdgcn_sched_barrier(0);
                wait_all(o);
                read_a<NT, NF, 1, 1, 1>(o, ab, af, lk);
                __builtin_amdgcn_sched_barrier(0);
                step_mfma<P, NF, 0, 1, 0>(acc, o);
                issue_range(D2{}, DN{});
                __builtin_amdgcn_sched_barrier(0);
                wait_all(o);
                publish();
                if (g + 1 < n) {
                    read_b<P, 0>(o, bb, lk);
                    read_a<NT, NF, 0, 0, 0>(o, ab, af, lk);
                }
                __builtin_amdgcn_sched_barrier(0);
                step_mfma<P, NF, 1, 1, 1>(acc, o);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                wait_all(o);
                read_b<P, 1>(o, bb, lk);
                read_a<NT, NF, 0, 1, 1>(o, ab, af, lk);
                __builtin_amdgcn_sched_barrier(0);
                step_mfma<P, NF, 0, 0, 0>(acc, o);
                issue_range(D0{}, DN{});
                __builtin_amdgcn_sched_barrier(0);
                wait_all(o);
                publish();
                if (g + 1 < n) {
                    read_b<P, 0>(o, bb, lk);
                    read_a<NT, NF, 0, 0, 0>(o, ab, af, lk);
                }
                __builtin_amdgcn_sched_barrier(0);
                step_mfma<P, NF, 0, 1, 1>(acc, o);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    int cb, cy0, cx0, cn0;
    for (int k = 0; k < my_tiles; ++k) {
        decode(k, cb, cy0, cx0, cn0);
        if (nch2) run(PhCentre{}, nch2);  // residual 1x1 segment
        run(PhCentre{}, nchC);            // sub-position 0: tap (0, 0)
        run(PhLeft{}, nchC);              // sub-position 1: taps (0, -1), (0, 0)
        run(PhUp{}, nchC);                // sub-position 2: taps (-1, 0), (0, 0)
        run(PhUpLeft{}, nchC);            // sub-position 3: taps (-1, -1), (-1, 0)
        run(PhLeft{}, nchC);              //                 taps (0, -1), (0, 0)
        if constexpr (NF == 4)
            asm volatile("s_nop 15\n\ts_nop 15"
                         : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                           "+v"(acc[1][2]), "+v"(acc[1][3]));
        else
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        epilogue(cb, cy0, cx0, cn0);
        zero_acc();
    }
    (void)stages_per_tile;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMAs must not outlive the block's LDS allocation
}

int g_sf_cus = 0;
unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

template <int TW, int NT, int NW, int NF>
int launch_sf(const ConvKParams& p, hipStream_t s) {
    using Cfg = FwCfg<TW, NT, NW, NF>;
    static int attr_state = 0;
    if (attr_state == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2d_fwd_kernel<TW, NT, NW, NF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr_state = e == hipSuccess ? 1 : -1;
    }
    if (attr_state < 0) return STYLEX_NOT_APPLICABLE;
    if (!g_sf_cus) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        g_sf_cus = n > 0 ? (n & ~7) : 256;
        if (g_sf_cus < 8) g_sf_cus = 8;
    }
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / Cfg::TH, ngs = p.N / NT;
    SfArgs sa;
    sa.total_tiles = p.B * tiles_x * tiles_y * ngs;
    sa.m_ng = magic_of(ngs);
    sa.m_tpi = magic_of(tiles_x * tiles_y);
    sa.m_tx = magic_of(tiles_x);
    stylex_note_kernel("conv_s2d_fwd_kernel<%d, %d, %d, %d>", TW, NT, NW, NF);
    const int blocks = NW == 4 ? 2 * g_sf_cus : g_sf_cus;  // 4-wave blocks: two per CU
    hipLaunchKernelGGL((conv_s2d_fwd_kernel<TW, NT, NW, NF>), dim3((unsigned)blocks), dim3(NW * 64), Cfg::SMEM, s, p, sa);
    return (int)hipGetLastError();
}

}  // namespace

// forward of the space-to-depth stride-2 conv: bf16, bias / residual merge epilogue, optional 1x1 residual K segment, whole
// tiles (8 x 32 or 16 x 16 pixels x 256 channels, 16 x 32 x 128), 32-channel K stages
int stylex_launch_s2d_fwd(const ConvKParams& p, hipStream_t s) {
    const char* env = getenv("STYLEX_S2D_FWD");  // read per launch: A/B tests toggle it in-process
    if (env && env[0] == '0') return STYLEX_NOT_APPLICABLE;
    if (!p.act_bf16 || !p.s2d_c || p.flip_taps || p.a_scale || p.mask || p.gate_mask) return STYLEX_NOT_APPLICABLE;
    if (p.flags & ~(STYLEX_EPI_BIAS | STYLEX_EPI_RESIDUAL)) return STYLEX_NOT_APPLICABLE;
    if (p.Ck != 4 * p.s2d_c || p.s2d_c % 32 != 0 || p.N % 64 != 0) return STYLEX_NOT_APPLICABLE;
    if ((p.flags & STYLEX_EPI_BIAS) && (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15))) return STYLEX_NOT_APPLICABLE;
    if ((p.flags & STYLEX_EPI_RESIDUAL) && (!p.residual || p.x2 || (reinterpret_cast<uintptr_t>(p.residual) & 7))) return STYLEX_NOT_APPLICABLE;
    if (p.x2) {
        if (!p.w2 || p.c2 < 8 || p.c2 % 8 != 0) return STYLEX_NOT_APPLICABLE;
        if ((reinterpret_cast<uintptr_t>(p.x2) & 15) || (reinterpret_cast<uintptr_t>(p.w2) & 15)) return STYLEX_NOT_APPLICABLE;
        if ((long)p.B * p.Ho * p.Wo * p.c2 * 2 >= (1l << 30)) return STYLEX_NOT_APPLICABLE;
    }
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (reinterpret_cast<uintptr_t>(p.y) & 15))
        return STYLEX_NOT_APPLICABLE;
    if ((long)(p.B * p.Ho + 2) * p.Wo * p.Ck * 2 >= (1l << 31) - 16 || (long)p.B * p.Ho * p.Wo * p.N * 2 >= (1l << 31) - 16 ||
        (long)p.N * 9 * p.Ck * 2 >= (1l << 30))
        return STYLEX_NOT_APPLICABLE;
    const bool n256 = p.N % 256 == 0;
    const bool w32 = p.Wo % 32 == 0 && p.Ho % 8 == 0, w16 = p.Wo % 16 == 0 && p.Ho % 16 == 0;
    if (!w32 && !w16) return STYLEX_NOT_APPLICABLE;
    if (p.dry) return 0;
    if (p.N % 128 != 0)  // 64-channel layers: 256 px x 64 n tiles in 4-wave blocks (HBM-side work: 0.67 GB per 77 GFLOP at B = 64)
        return w32 ? launch_sf<32, 64, 4, 2>(p, s) : launch_sf<16, 64, 4, 2>(p, s);
    // Default: 256 px x 128 n tiles in 4-wave blocks, two blocks per CU — measured equal to or faster than the 8-wave blocks
    // with 256 x 256 / 512 x 128 tiles on every layer and batch but one (256 -> 256 @64^2, B = 128: 0.176 vs 0.170 ms): the
    // second resident block covers the other's barriers, phase starts and epilogue, and small launches spread over twice
    // the tiles (profiles/r05_s_s2d_fwd_tiles.txt).  STYLEX_S2D_FWD_TILE=1 selects the 8-wave blocks with
    // the wide tiles, =2 the 4-wave blocks whatever the tile count.
    const char* te = getenv("STYLEX_S2D_FWD_TILE");
    if (te && te[0] == '1' && (n256 || (w32 && p.Ho % 16 == 0))) {
        if (n256) return w32 ? launch_sf<32, 256, 8, 4>(p, s) : launch_sf<16, 256, 8, 4>(p, s);
        return launch_sf<32, 128, 8, 4>(p, s);
    }
    // at most one tile per CU: the same tile on 8 waves (two per SIMD cover each other's waits; a lone 4-wave block is
    // latency-bound at 1.3 us per stage)
    const long tiles = (long)p.B * p.Ho * p.Wo / 256 * (p.N / 128);
    if (!(te && te[0] == '2') && tiles <= 256) return w32 ? launch_sf<32, 128, 8, 2>(p, s) : launch_sf<16, 128, 8, 2>(p, s);
    return w32 ? launch_sf<32, 128, 4, 4>(p, s) : launch_sf<16, 128, 4, 4>(p, s);
}
