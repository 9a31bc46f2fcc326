// conv_wgrad_pipe.hip — round 5: weight gradient of the 3x3 / stride-1 / pad-1 bf16 layers (discriminator / encoder
// convs, /root/reference/stylex/stylex_train.py:724-736; modulated generator convs :647-667 through the x scale) on the
// recipe of conv_pipe.hip: one 8-wave block per CU, LDS-DMA staging through rings that are never drained, counted
// waits, hand-placed LDS transpose reads.
//
//   dW[n][tap][c] = sum_{b,y,x} dy[b,y,x,n] * xin[b,y+kh-1,x+kw-1,c]          (GEMM: D[n][c] += A[n][k=pixel] B[k][c])
//
// What was wrong with conv3x3_wgrad_halo_dma_kernel (round-4 counters: 41-45 % of the wave cycles in waits, 2.5x the
// LDS instructions per MFMA of the forward kernel, a vmcnt(0) + __syncthreads per tile):
//   * every (tap, k-step) fetched its own B fragment: 2 + 18 transpose reads per 9 MFMAs.  The x fragment of tap
//     (kh, kw) at output row r is the fragment of tap (kh-1, kw) at row r+1 — it depends on the halo row rho = r + kh
//     only.  A wave now walks its tile by HALO ROW: the three kw fragments of row rho are read once and multiplied
//     with the dy fragments of rows rho, rho-1, rho-2 (kept in a four-deep register ring) — 88 reads per 72 MFMAs
//     instead of 160;
//   * 64-byte DMA pieces of 32-channel panels: the stage is now whole 128-byte lines (64 channels per pixel row, the
//     two 64-byte halves swapped when bit 1 of the pixel index is set, so that the four rows of a transpose read fall
//     on four different bank quarters);
//   * a 128(n) x 64(c) block tile (the dy stage is shared by two c halves, the x halo by four n blocks): 58 KB
//     staged per 37.7 MFLOP instead of 75.5 KB per 18.9;
//   * x one stage ahead, dy two: the wait in front of a stage's barrier is `vmcnt(#dy pieces)`, the DMA of the next
//     stages is issued two pieces per step behind the first MFMAs of a stage (a piece issued late in a stage is
//     waited for a few hundred cycles later — the first version, one piece per step, stalled every stage on its last
//     piece), and the barrier sits in front of the LAST step's MFMAs so that the first operands of the next stage are
//     already on their way.
// Layers with 64 output channels (HBM-side: 64 -> 64 @256^2 streams 2 x 537 MB at B = 64) use the same wave code on a
// 64(n) x 64(c) tile: the two wave quartets split the stage's pixels (k-step columns) and both streams run two stages
// ahead (3 x 26 + 3 x 16 KB); the quartets' accumulators are added through LDS before the block writes ONE partial
// slice.  Partials: workspace[slice][n][tap][c], reduced in fixed order by wgrad_reduce_kernel (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// kernel argument (outside the anonymous namespace: a kernel's host stub needs externally visible parameter types)
struct StylexWgPipeArgs {
    int total_tiles;      // B * TX * TY stages of the launch
    int tiles_per_split;  // stages per block
    int otiles, c_tiles;  // output tiles (n tiles x c tiles), c tiles
    int tx_count, ty_count;
};

namespace {

typedef StylexWgPipeArgs WgArgs;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// NP64 = 64-channel dy panels per stage (block tile 64*NP64 output channels), TW = tile width in pixels.
// A stage is PX output pixels (TR rows x TW) with its (TR+2) x (TW+2) halo of x; every wave multiplies 128 of them.
template <int NP64, int TW>
struct WgCfg {
    static constexpr int kTW = TW;
    static constexpr int PX = 128;                       // output pixels per stage
    static constexpr int TR = PX / TW;                   // 4 / 8 rows
    static constexpr int HWD = TW + 2;
    static constexpr int XPX = (TR + 2) * HWD;           // 204 (TW 32), 180 (TW 16)
    static constexpr int X_PIECES = (XPX + 7) / 8;       // 1 KiB DMA pieces = 8 pixel rows of 128 bytes
    static constexpr int X_STRIDE = X_PIECES * 1024;
    static constexpr int DY_PIECES = 16 * NP64;
    static constexpr int DY_STRIDE = DY_PIECES * 1024;   // NP64 panels x 128 pixels x 128 bytes
    static constexpr int XA = NP64 == 2 ? 1 : 2, DA = 2; // stages the x / dy streams run ahead of the consumer
    static constexpr int NX = XA + 1, ND = DA + 1;       // ring depths
    static constexpr int XS = (X_PIECES + 7) / 8, DS = DY_PIECES / 8, NSLOT = XS + DS;  // DMA instructions per wave and stage
    static constexpr int WAITN = (XA == 2 ? XS : 0) + (DA == 2 ? DS : 0);              // pieces that may stay in flight
    static constexpr int DY_BASE = NX * X_STRIDE;
    static constexpr int DUMP_BASE = DY_BASE + ND * DY_STRIDE;  // 1 KiB: destination of the pieces a wave does not have
    static constexpr int COMBINE = NP64 == 1 ? 4 * 5 * 4096 : 0;  // quartet combine: 4 wave pairs x 5 accumulator tiles
    static constexpr int SMEM = DUMP_BASE + 1024 > COMBINE ? DUMP_BASE + 1024 : COMBINE;  // <= 149 KiB: one block per CU
    // NP64 == 2: a wave multiplies the whole stage (RW rows x PWS 16-pixel columns); NP64 == 1: half of it — one of
    // the two k-step columns (TW 32) or four of the eight rows (TW 16)
    static constexpr int RW = NP64 == 2 ? TR : 4;
    static constexpr int PWS = (NP64 == 2 && TW == 32) ? 2 : 1;
    static constexpr int NSTEP = PWS * (RW + 2);          // steps (halo row, k-step column) per stage: 12 / 10 / 6
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* smem, int lds_off, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(smem + lds_off), 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS transpose read (4 consecutive pixels of this lane's channel) as inline asm: hipcc puts vmcnt(0) in front of every
// LDS read it can see behind a buffer-load-to-LDS and sinks the reads next to their MFMAs (conv_pipe.hip).
template <int OFF>
__device__ __forceinline__ void tr_read(s16x4& dst, int addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

__device__ __forceinline__ void mfma1(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ bf16x8 cat8(const s16x4& lo, const s16x4& hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Operand registers of a wave: B = the three kw fragments of one halo row (ping-pong by step parity), A = dy fragments
// of four output rows (ring by row).  A fragment is two transpose reads (pixels +0..3 and +4..7 of the lane's k half).
struct WgOps {
    s16x4 b[2][3][2];
    s16x4 a[4][2];
};

// step S of a stage: halo row rho = S % (RW+2), k-step column pw = S / (RW+2)
// MASK = the taps (bit kh * 3 + kw) this block multiplies: all nine, or the structurally non-zero taps of one
// sub-position of a space-to-depth stride-2 conv (stylex_s2d_tap_mask)
constexpr unsigned WG_ALL_TAPS = 0x1ffu;
constexpr bool wg_kw_used(unsigned mask, int kw) { return ((mask >> kw) | (mask >> (3 + kw)) | (mask >> (6 + kw))) & 1u; }

template <class Cfg, int S, unsigned MASK = WG_ALL_TAPS>
__device__ __forceinline__ void issue_reads(WgOps& o, const int (&xb)[4], int dyb) {
    constexpr int rho = S % (Cfg::RW + 2), pw = S / (Cfg::RW + 2), par = S & 1;  // pw < PWS
    // the swizzled half of a pixel row depends on bits 1:0 of its index: (rho * HWD + kw) % 4, and HWD % 4 == 2
    static_assert(Cfg::HWD % 4 == 2, "swizzle class of a halo pixel");
    constexpr int q0 = (rho * Cfg::HWD + pw * 16 + 0) * 128, q1 = (rho * Cfg::HWD + pw * 16 + 1) * 128,
                  q2 = (rho * Cfg::HWD + pw * 16 + 2) * 128;
    constexpr int m0 = (2 * rho + 0) & 3, m1 = (2 * rho + 1) & 3, m2 = (2 * rho + 2) & 3;
    if constexpr (wg_kw_used(MASK, 0)) {
        tr_read<q0>(o.b[par][0][0], xb[m0]);
        tr_read<q0 + 512>(o.b[par][0][1], xb[m0]);
    }
    if constexpr (wg_kw_used(MASK, 1)) {
        tr_read<q1>(o.b[par][1][0], xb[m1]);
        tr_read<q1 + 512>(o.b[par][1][1], xb[m1]);
    }
    if constexpr (wg_kw_used(MASK, 2)) {
        tr_read<q2>(o.b[par][2][0], xb[m2]);
        tr_read<q2 + 512>(o.b[par][2][1], xb[m2]);
    }
    if constexpr (rho < Cfg::RW) {
        constexpr int qa = (rho * Cfg::kTW + pw * 16) * 128;
        tr_read<qa>(o.a[rho & 3][0], dyb);
        tr_read<qa + 512>(o.a[rho & 3][1], dyb);
    }
}

template <class Cfg, int S>
__device__ __forceinline__ void wait_reads(WgOps& o) {
    constexpr int rho = S % (Cfg::RW + 2), par = S & 1;
    if constexpr (rho < Cfg::RW)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(o.b[par][0][0]), "+v"(o.b[par][0][1]), "+v"(o.b[par][1][0]), "+v"(o.b[par][1][1]),
                       "+v"(o.b[par][2][0]), "+v"(o.b[par][2][1]), "+v"(o.a[rho & 3][0]), "+v"(o.a[rho & 3][1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(o.b[par][0][0]), "+v"(o.b[par][0][1]), "+v"(o.b[par][1][0]), "+v"(o.b[par][1][1]),
                       "+v"(o.b[par][2][0]), "+v"(o.b[par][2][1]));
}

// MFMAs of step S: x row rho serves tap row kh of output row r = rho - kh.  BIAS: one more MFMA per new dy fragment,
// dy x ONES into accb — every column of accb = sum over the pixels of dy[., n] (the bias gradient).
template <class Cfg, int S, bool BIAS, unsigned MASK = WG_ALL_TAPS>
__device__ __forceinline__ void step_mfma(f32x16 (&acc)[9], f32x16& accb, const WgOps& o, bool do_bias, const bf16x8& ones) {
    constexpr int rho = S % (Cfg::RW + 2), par = S & 1;
    const bf16x8 b0 = cat8(o.b[par][0][0], o.b[par][0][1]);
    const bf16x8 b1 = cat8(o.b[par][1][0], o.b[par][1][1]);
    const bf16x8 b2 = cat8(o.b[par][2][0], o.b[par][2][1]);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int r = rho - kh;
        if (r >= 0 && r < Cfg::RW) {
            const bf16x8 av = cat8(o.a[r & 3][0], o.a[r & 3][1]);
            if ((MASK >> (kh * 3 + 0)) & 1u) mfma1(acc[kh * 3 + 0], av, b0);
            if ((MASK >> (kh * 3 + 1)) & 1u) mfma1(acc[kh * 3 + 1], av, b1);
            if ((MASK >> (kh * 3 + 2)) & 1u) mfma1(acc[kh * 3 + 2], av, b2);
        }
    }
    if constexpr (BIAS && rho < Cfg::RW) {
        if (do_bias) {  // wave-uniform: a scalar branch around one MFMA
            const bf16x8 av = cat8(o.a[rho & 3][0], o.a[rho & 3][1]);
            mfma1(accb, av, ones);
        }
    }
}

constexpr unsigned WG_OOB = 0x80000000u;  // beyond num_records of every descriptor below: the load returns zeros

// uniform (SGPR) state of one DMA stream: the tile it is about to fetch
struct WgCursor {
    int t, b, tx, ty;
};

// S2D: the 3x3 / stride-2 conv in its space-to-depth form (a 3x3 / s1 conv over 4 x s2d_c channels, stylex_internal.h): a
// 64-channel tile lies in ONE sub-position, whose 1, 2 or 4 structurally non-zero taps are all the block multiplies; its
// partial goes straight to the FOLDED layout [n][original tap][c] of the stride-2 weight (no dW2, no fold launch).
template <int NP64, int TW, bool BIAS, bool S2D = false>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_pipe_kernel(ConvKParams p, WgArgs wa) {
    using Cfg = WgCfg<NP64, TW>;
    constexpr int TR = Cfg::TR, HWD = Cfg::HWD, XPX = Cfg::XPX, X_PIECES = Cfg::X_PIECES;
    constexpr int X_STRIDE = Cfg::X_STRIDE, DY_STRIDE = Cfg::DY_STRIDE, DY_BASE = Cfg::DY_BASE;
    constexpr int XS = Cfg::XS, DS = Cfg::DS, NSLOT = Cfg::NSLOT, NSTEP = Cfg::NSTEP, NX = Cfg::NX, ND = Cfg::ND;
    constexpr int XA = Cfg::XA, DA = Cfg::DA, WAITN = Cfg::WAITN;
    static_assert(NSTEP % 2 == 0, "the operand ping-pong restarts at parity 0 every stage");
    static_assert(NSLOT <= 2 * (NSTEP - 1), "at most two DMA pieces per step");
    static_assert(XA <= DA, "issue order inside a stage: the one-ahead stream first");
    static_assert(!(S2D && BIAS), "the space-to-depth path has no bias sums");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.Wo, C = p.Ck, N = p.N;

    // XCD-aware order (as conv_wgrad_halo.hip): the blocks of one split — they stream the same pixels, each for its own
    // channel tile — get consecutive logical ids on ONE XCD, so the re-reads hit that XCD's L2
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, rr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    // integer divisions run on the VALU: pin their (uniform) results to SGPRs, or every DMA instruction whose scalar
    // offset descends from them is wrapped in a waterfall loop
    const int split = __builtin_amdgcn_readfirstlane(bid / wa.otiles);
    const int ot = bid - split * wa.otiles;
    const int nt_ = __builtin_amdgcn_readfirstlane(ot / wa.c_tiles);
    const int n0 = nt_ * (64 * NP64), c0 = (ot - nt_ * wa.c_tiles) * 64;
    const int t_begin = split * wa.tiles_per_split;
    const int t_end = min(wa.total_tiles, t_begin + wa.tiles_per_split);
    const int nst = t_end - t_begin;
    if (nst <= 0) return;

    // ---- wave roles: channel block of x (cblk), 32-channel half of a dy panel (nhalf), and hi2 = the dy panel
    // (NP64 == 2) or the half of the stage's pixels this quartet multiplies (NP64 == 1)
    const int cblk = wave & 1, nb4 = wave >> 1;
    const int nhalf = nb4 & 1, hi2 = nb4 >> 1;
    const int dy_wave_off = NP64 == 2 ? hi2 * 16384 : (TW == 32 ? hi2 * 16 * 128 : hi2 * 4 * TW * 128);
    const int x_wave_off = NP64 == 2 ? 0 : (TW == 32 ? hi2 * 16 * 128 : hi2 * 4 * HWD * 128);  // multiples of 4 pixels

    // ---- transpose-read lane addressing: 16-lane group g -> channel block (g & 1) * 16, k half (g >> 1) * 8 pixels;
    // lane i of a group supplies pixel i >> 2, channels 4 * (i & 3)
    const int i16 = lane & 15, g = lane >> 4;
    const int lane_px = ((g >> 1) * 8 + (i16 >> 2)) * 128 + (g & 1) * 32 + (i16 & 3) * 8;
    int dyb = lane_px + ((nhalf ^ ((i16 >> 3) & 1)) * 64) + dy_wave_off + DY_BASE;
    int xb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) xb[m] = lane_px + ((cblk ^ (((m + (i16 >> 2)) >> 1) & 1)) * 64) + x_wave_off;

    // ---- DMA lane constants.  A piece = 8 pixel rows x 128 bytes; lane -> pixel lane >> 3, physical 16-byte slot
    // lane & 7, which holds the logical slot with the 64-byte halves swapped when bit 1 of the pixel index is set.
    const int dpx = lane >> 3;
    const unsigned lslot = (unsigned)((lane & 7) ^ (((lane >> 4) & 1) << 2));
    // x: pixel q of the halo = (row, col); voffset relative to (y0 - 1, x0 - 1) — the descriptor's base is moved back by
    // one row + one pixel so that it is never negative; bits 3:0 = (top row, bottom row, left column, right column)
    unsigned vxc[XS];
    int xdst[XS];
#pragma unroll
    for (int k = 0; k < XS; ++k) {
        const int piece = wave + 8 * k;
        const int q = piece * 8 + dpx;
        const int row = q / HWD, col = q - row * HWD;
        const bool ok = piece < X_PIECES && q < XPX && c0 + (int)lslot * 8 < C;  // C == 32: half of the 64-channel row
        const unsigned edge = (row == 0 ? 1u : 0u) | (row == TR + 1 ? 2u : 0u) | (col == 0 ? 4u : 0u) | (col == TW + 1 ? 8u : 0u);
        vxc[k] = ok ? (((unsigned)(row * W + col) * (unsigned)C + (unsigned)c0 + lslot * 8u) * 2u) | edge : WG_OOB;
        xdst[k] = piece < X_PIECES ? piece * 1024 : -1;
    }
    unsigned vdy[DS];
#pragma unroll
    for (int k = 0; k < DS; ++k) {
        const int piece = wave + 8 * k;  // panel = piece >> 4
        const int q = (piece & 15) * 8 + dpx;
        const int r = q / TW, col = q - r * TW;
        const int nch = n0 + (piece >> 4) * 64 + (int)lslot * 8;
        vdy[k] = nch < N ? ((unsigned)(r * W + col) * (unsigned)N + (unsigned)nch) * 2u : WG_OOB;  // N == 32: half rows
    }
    // (the lane-constant code above leaves W in a VGPR behind a divergent branch: scalar copies for everything below)
    const int Hs = __builtin_amdgcn_readfirstlane(p.Ho), Ws = __builtin_amdgcn_readfirstlane(p.Wo);
    const int Cs = __builtin_amdgcn_readfirstlane(p.Ck), Ns = __builtin_amdgcn_readfirstlane(p.N);
    const uint64_t xbase = reinterpret_cast<uint64_t>(p.a) - (uint64_t)((long)(Ws + 1) * Cs * 2);
    const uint64_t xbase_u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(xbase >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)(xbase & 0xffffffffu));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(xbase_u), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a2), 0, 0x40000000, 0x00020000);

    // ---- tile cursors of the two DMA streams (uniform): tiles are ordered (b, tx, ty) with ty fastest — a block walks
    // down a 32-pixel column, so the two halo rows a tile shares with the one above were fetched one stage ago (L2)
    WgCursor cx, cd;
    {
        const int r1 = __builtin_amdgcn_readfirstlane(t_begin / wa.ty_count);
        const int b1 = __builtin_amdgcn_readfirstlane(r1 / wa.tx_count);
        cx.t = t_begin, cx.ty = t_begin - r1 * wa.ty_count, cx.tx = r1 - b1 * wa.tx_count, cx.b = b1;
        cd = cx;
    }
    auto advance = [&](WgCursor& c) {
        ++c.t;
        if (++c.ty == wa.ty_count) {
            c.ty = 0;
            if (++c.tx == wa.tx_count) c.tx = 0, ++c.b;
        }
    };
    int xslot_dma = 0, dslot_dma = 0;  // ring slots the streams write next
    // (plain macros, not lambdas: a lambda that captures a buffer resource makes the HOST pass drop the kernel's stub)
#define WG_ISSUE_X(K)                                                                                                \
    {                                                                                                                \
        const bool more_ = cx.t < t_end;                                                                             \
        const unsigned soff_ = (unsigned)(((cx.b * Hs + cx.ty * TR) * Ws + cx.tx * TW) * Cs) * 2u;                   \
        const unsigned edge_ = (cx.ty == 0 ? 1u : 0u) | (cx.ty == wa.ty_count - 1 ? 2u : 0u) | (cx.tx == 0 ? 4u : 0u) | \
                               (cx.tx == wa.tx_count - 1 ? 8u : 0u);                                                 \
        const unsigned v_ = ((vxc[K] & edge_) || !more_) ? WG_OOB : (vxc[K] & ~15u);                                 \
        dma16(rx, smem, xdst[K] >= 0 ? xslot_dma * X_STRIDE + xdst[K] : Cfg::DUMP_BASE, v_, more_ ? soff_ : 0u);     \
    }
#define WG_ISSUE_DY(K)                                                                                               \
    {                                                                                                                \
        const bool more_ = cd.t < t_end;                                                                             \
        const unsigned soff_ = (unsigned)(((cd.b * Hs + cd.ty * TR) * Ws + cd.tx * TW) * Ns) * 2u;                   \
        dma16(rdy, smem, DY_BASE + dslot_dma * DY_STRIDE + (wave + 8 * (K)) * 1024, more_ ? vdy[K] : WG_OOB,          \
              more_ ? soff_ : 0u);                                                                                   \
    }
#define WG_END_X()  { advance(cx); xslot_dma = xslot_dma == NX - 1 ? 0 : xslot_dma + 1; }
#define WG_END_DY() { advance(cd); dslot_dma = dslot_dma == ND - 1 ? 0 : dslot_dma + 1; }
    // slot K of a stage's NSLOT DMA instructions: the x pieces first, then the dy pieces (the counted wait relies on it)
#define WG_ISSUE(K)                                            \
    if constexpr ((K) < XS) {                                  \
        WG_ISSUE_X((K) < XS ? (K) : 0)                         \
        if constexpr ((K) == XS - 1) WG_END_X()                \
    } else if constexpr ((K) < NSLOT) {                        \
        WG_ISSUE_DY((K) - XS < DS ? ((K) >= XS ? (K) - XS : 0) : 0) \
        if constexpr ((K) == NSLOT - 1) WG_END_DY()            \
    }

    f32x16 acc[9], accb;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;
    const bool do_bias = BIAS && p.bias_partial != nullptr && c0 == 0 && cblk == 0;
    bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};  // bf16 1.0
    asm volatile("" : "+v"(ones));  // one register quad for the whole kernel, not a re-materialisation per step

    // ---- prologue: stage 0 of both streams, then stage 1 of the streams that run two ahead
#pragma unroll
    for (int k = 0; k < XS; ++k) WG_ISSUE_X(k)
    WG_END_X()
#pragma unroll
    for (int k = 0; k < DS; ++k) WG_ISSUE_DY(k)
    WG_END_DY()
    if constexpr (XA == 2) {
#pragma unroll
        for (int k = 0; k < XS; ++k) WG_ISSUE_X(k)
        WG_END_X()
    }
    if constexpr (DA == 2) {
#pragma unroll
        for (int k = 0; k < DS; ++k) WG_ISSUE_DY(k)
        WG_END_DY()
    }
    WgOps o{};  // (zero: a masked variant names fragments it never reads as wait operands)
    wait_vmcnt<WAITN>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int xslot_c = 0, dslot_c = 0;  // consumer's ring slots

    // one step: wait for this step's operands (requested one step ago), request the next step's, MFMAs, DMA issue
    // (two pieces per step from the first step on: everything a stage issues has most of the stage to land)
#define WG_STEP(S)                                                                     \
    wait_reads<Cfg, S>(o);                                                             \
    issue_reads<Cfg, (S) + 1, MASK_>(o, xb, dyb);                                      \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    step_mfma<Cfg, S, BIAS, MASK_>(acc, accb, o, do_bias, ones);                       \
    if constexpr (2 * (S) < NSLOT) { WG_ISSUE(2 * (S)) }                               \
    if constexpr (2 * (S) + 1 < NSLOT) { WG_ISSUE(2 * (S) + 1) }                       \
    __builtin_amdgcn_sched_barrier(0);
    // the last step: the barrier that publishes the next stage sits in front of its MFMAs
#define WG_LAST(S)                                                                     \
    wait_reads<Cfg, S>(o);                                                             \
    wait_vmcnt<WAITN>();                                                               \
    __builtin_amdgcn_s_barrier();                                                      \
    asm volatile("" ::: "memory");                                                     \
    {                                                                                  \
        const int dx_ = xslot_c == NX - 1 ? -(NX - 1) * X_STRIDE : X_STRIDE;           \
        const int dd_ = dslot_c == ND - 1 ? -(ND - 1) * DY_STRIDE : DY_STRIDE;         \
        xslot_c = xslot_c == NX - 1 ? 0 : xslot_c + 1;                                 \
        dslot_c = dslot_c == ND - 1 ? 0 : dslot_c + 1;                                 \
        _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) xb[m_] += dx_;                \
        dyb += dd_;                                                                    \
    }                                                                                  \
    issue_reads<Cfg, 0, MASK_>(o, xb, dyb);                                            \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    step_mfma<Cfg, S, BIAS, MASK_>(acc, accb, o, do_bias, ones);                       \
    __builtin_amdgcn_sched_barrier(0);
    // the stage loop for one tap set (a compile-time mask: a per-tap runtime branch between MFMAs makes every MFMA wait
    // for its own operand read, conv_wgrad_halo.hip)
#define WG_LOOP(M)                                        \
    {                                                     \
        constexpr unsigned MASK_ = (M);                   \
        issue_reads<Cfg, 0, MASK_>(o, xb, dyb);           \
        for (int j = 0; j < nst; ++j) {                   \
            WG_STEP(0)                                    \
            WG_STEP(1)                                    \
            WG_STEP(2)                                    \
            WG_STEP(3)                                    \
            WG_STEP(4)                                    \
            if constexpr (NSTEP == 6) {                   \
                WG_LAST(5)                                \
            } else {                                      \
                WG_STEP(5)                                \
                WG_STEP(6)                                \
                WG_STEP(7)                                \
                WG_STEP(8)                                \
                if constexpr (NSTEP == 12) {              \
                    WG_STEP(9)                            \
                    WG_STEP(10)                           \
                    WG_LAST(11)                           \
                } else {                                  \
                    WG_LAST(9)                            \
                }                                         \
            }                                             \
        }                                                 \
    }
    const int sub = S2D ? c0 / p.s2d_c : 0;  // sub-position (sy, sx) of this block's channel tile
    if constexpr (!S2D) {
        WG_LOOP(WG_ALL_TAPS)
    } else {
        if (sub == 0) WG_LOOP(0x010u)
        else if (sub == 1) WG_LOOP(0x018u)
        else if (sub == 2) WG_LOOP(0x012u)
        else WG_LOOP(0x01bu)
    }
#undef WG_LOOP
#undef WG_LAST
#undef WG_STEP
#undef WG_ISSUE
#undef WG_END_DY
#undef WG_END_X
#undef WG_ISSUE_DY
#undef WG_ISSUE_X
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the look-ahead operand reads of the (non-existent) next stage
    wait_vmcnt<0>();                                     // tail DMAs (zeros) must not outlive the block's LDS allocation

    // MFMA results -> VALU readers: 12+ wait states the compiler cannot see behind an asm MFMA
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                   "+v"(acc[7]), "+v"(acc[8]), "+v"(accb));

    // ---- NP64 == 1: the second quartet's accumulators (the other half of every stage's pixels) are added to the first
    // quartet's through LDS, five tiles per round (4 wave pairs x 5 x 4 KiB), in a fixed order
    if constexpr (NP64 == 1) {
        float4* cb = reinterpret_cast<float4*>(smem) + (wave & 3) * (5 * 4 * 64) + lane;
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            __syncthreads();  // round 0: every wave's operand reads and tail DMAs are done; round 1: round 0 was read
            if (hi2 == 1) {
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) {
                    const int t = round * 5 + tt;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x16& v = t < 9 ? acc[t < 9 ? t : 0] : accb;
                        cb[(tt * 4 + q) * 64] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    }
                }
            }
            __syncthreads();
            if (hi2 == 0) {
#pragma unroll
                for (int tt = 0; tt < 5; ++tt) {
                    const int t = round * 5 + tt;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 u = cb[(tt * 4 + q) * 64];
                        f32x16& v = t < 9 ? acc[t < 9 ? t : 0] : accb;
                        v[4 * q] += u.x, v[4 * q + 1] += u.y, v[4 * q + 2] += u.z, v[4 * q + 3] += u.w;
                    }
                }
            }
        }
        if (hi2 == 1) return;
    }

    // ---- partial[split][n][tap][c]; D[i = n][j = c]: col j = lane & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int nbase = n0 + (NP64 == 2 ? nb4 : nhalf) * 32;
    const int lj = lane & 31, lh = lane >> 5;
    if constexpr (S2D) {
        // folded layout: frame tap (kh2, kw2) of sub-position (sy, sx) is tap (kh, kw) of the stride-2 weight with
        // kh = kh2 == 0 ? 0 : sy + 1 (stylex_fold_weight_grad_s2d's map read backwards), channel c - sub * s2d_c
        const int Co = p.s2d_c, sy = sub >> 1, sx = sub & 1;
        float* outf = p.y + (long)split * N * 9 * Co;
        const int cl = c0 - sub * Co + cblk * 32 + lj;
#pragma unroll
        for (int t2 = 0; t2 < 5; ++t2) {
            if (t2 == 2) continue;  // frame taps 0, 1, 3, 4
            const int kh2 = t2 / 3, kw2 = t2 % 3;
            if ((kh2 == 0 && sy == 0) || (kw2 == 0 && sx == 0)) continue;  // structurally zero for this sub-position
            const int tap = (kh2 == 0 ? 0 : sy + 1) * 3 + (kw2 == 0 ? 0 : sx + 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                outf[((long)n * 9 + tap) * Co + cl] = acc[t2][r];
            }
        }
        return;
    }
    float* out = p.y + (long)split * N * 9 * C;
    const int c = c0 + cblk * 32 + lj;
    if (nbase >= N || c0 + cblk * 32 >= C) return;  // N == 32 / C == 32: this wave multiplied the zero half of a row
    if (do_bias && lj == 0) {  // every column of accb holds the same sums: column 0 writes them
#pragma unroll
        for (int r = 0; r < 16; ++r) p.bias_partial[(long)split * N + nbase + (r & 3) + 8 * (r >> 2) + 4 * lh] = accb[r];
    }
    // modulated layer: the whole split lies in one sample (plan), its x scale is a factor of the sum
    if (c >= C) return;  // C == 8: lanes 8 .. 31 of the first channel block multiplied zeros
    const float xsc = p.a_scale ? p.a_scale[(long)(t_begin / (wa.tx_count * wa.ty_count)) * C + c] : 1.f;  // cx.b has moved on
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((long)n * 9 + t) * C + c] = acc[t][r] * xsc;
        }
}

template <int NP64, int TW>
int launch_wg_s2d(const ConvKParams& p, const WgArgs& wa, int blocks, hipStream_t s) {
    using Cfg = WgCfg<NP64, TW>;
    static int attr_state = 0;
    if (attr_state == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_pipe_kernel<NP64, TW, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        attr_state = e == hipSuccess ? 1 : -1;
    }
    if (attr_state < 0) return STYLEX_NOT_APPLICABLE;
    stylex_note_kernel("conv3x3_wgrad_pipe_kernel<%d, %d, false, true>", NP64, TW);
    hipLaunchKernelGGL((conv3x3_wgrad_pipe_kernel<NP64, TW, false, true>), dim3(blocks), dim3(512), Cfg::SMEM, s, p, wa);
    return (int)hipGetLastError();
}

template <int NP64, int TW>
int launch_wg(const ConvKParams& p, const WgArgs& wa, int blocks, bool bias, hipStream_t s) {
    using Cfg = WgCfg<NP64, TW>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_pipe_kernel<NP64, TW, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        if (e != hipSuccess) return STYLEX_NOT_APPLICABLE;  // a device with less LDS: the older kernels serve the launch
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_pipe_kernel<NP64, TW, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        if (e != hipSuccess) return STYLEX_NOT_APPLICABLE;
        attr_done = true;
    }
    stylex_note_kernel("conv3x3_wgrad_pipe_kernel<%d, %d, %s, false>", NP64, TW, bias ? "true" : "false");
    if (bias) hipLaunchKernelGGL((conv3x3_wgrad_pipe_kernel<NP64, TW, true>), dim3(blocks), dim3(512), Cfg::SMEM, s, p, wa);
    else hipLaunchKernelGGL((conv3x3_wgrad_pipe_kernel<NP64, TW, false>), dim3(blocks), dim3(512), Cfg::SMEM, s, p, wa);
    return (int)hipGetLastError();
}

int g_wg_cus = 0;

}  // namespace

// 64 (N % 128 != 0, or the launch is short: see the plan) or 128 output channels per block tile
static int wg_np64(const ConvKParams& p) {
    const char* env = getenv("STYLEX_WGRAD_PIPE_NP");  // A/B switch: 1 / 2 force the tile
    if (p.N % 128 != 0 || (env && env[0] == '1')) return 1;
    if (env && env[0] == '2') return 2;
    // few stages per block: the partial slices (one 128 x 64 x 9 fp32 tile = 295 KB per block, written once and read once
    // by the reduce launch) cost as much as the multiplications; the 64 x 64 tile halves them.  Stages per block at 128
    // channels = total stages / (CUs / output tiles); measured (tools/bench_wgrad.py --np-ab): 64-channel tiles are 1.03-1.16x
    // faster at <= 32 stages per block, equal from 64 on.
    const int tw = p.Wo >= 32 ? 32 : 16;
    const long tiles = (long)p.B * (p.Wo / tw) * (p.Ho / (128 / tw));
    const long otiles2 = (long)(p.N / 128) * ((p.Ck + 63) / 64);
    const long cus = g_wg_cus > 0 ? g_wg_cus : 256;
    const long per_block = tiles * otiles2 / cus;
    static const long thr = getenv("STYLEX_WGRAD_PIPE_NP_THR") ? atol(getenv("STYLEX_WGRAD_PIPE_NP_THR")) : 48;
    return per_block < thr ? 1 : 2;
}

// bf16 activations, 3x3 / s1 / p1, 64-channel tiles on both sides (or exactly 32 channels: the generator's last block),
// image = whole tiles, no dy scale, tensors below 1.5 GiB (32-bit buffer offsets)
bool stylex_wgrad_pipe_applicable(const ConvKParams& p) {
    const char* env = getenv("STYLEX_WGRAD_PIPE");  // read per launch: A/B tests toggle it in-process
    if (env && env[0] == '0') return false;
    if (!p.act_bf16 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.Hi != p.Ho || p.Wi != p.Wo) return false;
    if (p.a2_scale) return false;
    if (p.s2d_c && (p.Ck != 4 * p.s2d_c || p.s2d_c % 64 != 0 || p.N % 64 != 0 || p.a_scale)) return false;
    // input channels: whole 64-channel tiles, or exactly 32 (the generator's last block).  The padded RGB input of the
    // first conv (C = 8: one 16-byte slot of the row) runs correctly here as well but no faster than the flattened-tap
    // kernel of conv_wgrad_tr.hip (0.212 vs 0.212 ms at B = 64, 0.417 vs 0.402 at 128: profiles/r05_l_wgrad_ab.txt), which keeps it
    if ((p.Ck % 64 != 0 && p.Ck != 32) || (p.N % 64 != 0 && p.N != 32)) return false;
    const int tw = p.Wo >= 32 ? 32 : 16;
    if (p.Wo % tw != 0 || p.Wo < 16) return false;
    const int tr = 128 / tw;
    if (p.Ho % tr != 0) return false;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.a2) & 15)) return false;
    if (p.a_scale && (reinterpret_cast<uintptr_t>(p.a_scale) & 3)) return false;
    if ((long)p.B * p.Ho * p.Wo * p.Ck * 2 >= (1l << 30) + (1l << 29) || (long)p.B * p.Ho * p.Wo * p.N * 2 >= (1l << 30) + (1l << 29))
        return false;
    return true;
}

// slices = partial slices the launch writes (what the reduce kernel sums, what the workspace must hold)
void stylex_wgrad_pipe_plan(const ConvKParams& p, int* slices, int* tiles_per_split, int* blocks) {
    if (!g_wg_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        g_wg_cus = n > 0 ? n : 256;
    }
    const int np64 = wg_np64(p);
    const int tw = p.Wo >= 32 ? 32 : 16;
    const int tr = 128 / tw;
    const long tiles_img = (long)(p.Wo / tw) * (p.Ho / tr);
    const long tiles = (long)p.B * tiles_img;
    const long otiles = (long)((p.N + 64 * np64 - 1) / (64 * np64)) * ((p.Ck + 63) / 64);
    long want = g_wg_cus / otiles;
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    long tps = (tiles + want - 1) / want;
    if (p.a_scale) {  // splits must not cross a sample: the largest divisor of tiles-per-image that is <= tps
        if (tps > tiles_img) tps = tiles_img;
        while (tiles_img % tps) --tps;
    }
    const long splits = (tiles + tps - 1) / tps;
    *tiles_per_split = (int)tps;
    *blocks = (int)(splits * otiles);
    *slices = (int)splits;
}

int stylex_launch_wgrad_pipe(ConvKParams p, float* partial, hipStream_t s, int* slices_out, int* bias_done) {
    if (!stylex_wgrad_pipe_applicable(p)) return STYLEX_NOT_APPLICABLE;
    int slices, tps, blocks;
    stylex_wgrad_pipe_plan(p, &slices, &tps, &blocks);
    const int np64 = wg_np64(p);
    const int tw = p.Wo >= 32 ? 32 : 16;
    const int tr = 128 / tw;
    WgArgs wa;
    wa.tx_count = p.Wo / tw;
    wa.ty_count = p.Ho / tr;
    wa.total_tiles = p.B * wa.tx_count * wa.ty_count;
    wa.tiles_per_split = tps;
    wa.c_tiles = (p.Ck + 63) / 64;
    wa.otiles = ((p.N + 64 * np64 - 1) / (64 * np64)) * wa.c_tiles;
    p.y = partial;
    *slices_out = slices;
    if (p.s2d_c) {  // the partial / result is the FOLDED [N][9][s2d_c] gradient (stylex_launch_wgrad folds nothing more)
        if (bias_done) *bias_done = 0;
        p.bias_partial = nullptr;
        if (np64 == 2) return tw == 32 ? launch_wg_s2d<2, 32>(p, wa, blocks, s) : launch_wg_s2d<2, 16>(p, wa, blocks, s);
        return tw == 32 ? launch_wg_s2d<1, 32>(p, wa, blocks, s) : launch_wg_s2d<1, 16>(p, wa, blocks, s);
    }
    const bool bias = p.bias_partial != nullptr;
    if (bias_done) *bias_done = bias ? 1 : 0;
    if (np64 == 2) return tw == 32 ? launch_wg<2, 32>(p, wa, blocks, bias, s) : launch_wg<2, 16>(p, wa, blocks, bias, s);
    return tw == 32 ? launch_wg<1, 32>(p, wa, blocks, bias, s) : launch_wg<1, 16>(p, wa, blocks, bias, s);
}
