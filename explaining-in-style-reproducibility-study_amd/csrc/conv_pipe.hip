// conv_pipe.hip — persistent, software-pipelined LDS-DMA 3x3 / stride-1 / pad-1 forward and data gradient for the
// unmodulated (discriminator / encoder) layers, bf16 activations.  Round 3: replaces conv3x3_halo_dma_kernel<2> on the
// shapes it covers (reference: DiscriminatorBlock.net, /root/reference/stylex/stylex_train.py:724-736).
//
// Why.  The per-tile kernel (conv_halo_dma.hip) runs two 4-wave blocks per CU, each staging 38 KB (20 KB of halo + 18 KB
// of weights) per 16-channel chunk for 9.4 MFLOP, draining its DMA queue (vmcnt(0) + barrier) at every chunk and paying a
// prologue (first chunk's latency) and an epilogue per tile; its own ablation showed the DMA, MFMA and epilogue phases
// ADDING (.19 + .21 + .10 ms at 64->64 @256^2) instead of overlapping.  Here ONE 8-wave block per CU (still two waves per
// SIMD) owns a 512 px x 128 n (or 1024 px x 64 n) output tile:
//   * 56 KB staged per 18.9 MFLOP — 26 % fewer DMA bytes per FLOP (the halo is shared by the two channel halves /
//     the weights by twice the pixels);
//   * the block is persistent: it walks a static tile list (XCD-contiguous, channel tiles of a pixel tile adjacent) and
//     the chunk stream runs ACROSS tile boundaries — no prologue bubble per tile, the deferred epilogue of tile t is
//     issued after tile t+1's first operands are already in LDS;
//   * the halo ring has three stages, the weight ring two; the DMA queue is never drained: the wait before a chunk's
//     barrier is `s_waitcnt vmcnt(#halo pieces this wave issued last)`, which leaves the halo of chunk g+1 in flight
//     across the barrier of chunk g (loads retire in order, so everything older — the operands of chunk g and the
//     previous tile's stores — has landed);
//   * staging uses buffer_load ... lds with a wave-uniform SGPR offset for (tile, chunk) and a per-lane VGPR offset that
//     is out of range for padding pixels (the buffer bounds check returns zeros — no zero page, no 64-bit pointer math,
//     no divergent branches around the DMA instructions), one piece issued per tap in the shadow of that tap's MFMAs.
//
// LDS image (as conv_halo_dma.hip): 32-byte rows (16 bf16 channels); halo rows [NP], weight rows [9 taps][NT]; a row's
// two 16-byte halves are swapped when bit 3 of the row index is set (bank-conflict-free operand reads).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

// cache policy of the output stores (buffer_store aux bits; 2 = nt, streaming: tools/bench_s2d_dgrad.py A/B, DESIGN §3 "Round 5")
#ifndef PIPE_STORE_AUX
#define PIPE_STORE_AUX 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// kernel argument (not in the anonymous namespace: a kernel's host stub needs externally visible parameter types)
struct StylexPipeArgs {
    int total_tiles;
    unsigned m_ntiles, m_tpi, m_tx;  // magic reciprocals of n_tiles, tiles per image, tiles_x
    int dbg;                         // ablation switches of tools/bench_pipe.py (STYLEX_PIPE_DBG): 1 no stores, 2 every halo from tile 0, 4 every weight tile from n0 = 0, 8 do not wait for the epilogue stores, 16 store full 128-byte lines (8 and 16: WRONG results, timing ablations only)
};

// The ablation switches (some of which give WRONG results by design) exist only in a -DSTYLEX_PIPE_ABLATION build
// (`make ABLATION=1`, used by tools/bench_pipe.py); the production kernel compiles them out.
#ifdef STYLEX_PIPE_ABLATION
#define PIPE_DBG (pa.dbg)
#else
#define PIPE_DBG 0
#endif


namespace {

typedef StylexPipeArgs PipeArgs;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__device__ __forceinline__ unsigned short to_bf16(float v) {
    f32x2_t t = {v, 0.f};
    bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
    return (unsigned short)(*reinterpret_cast<unsigned*>(&r) & 0xffffu);
}

// NT = output channels per block tile.  128: 16x32 px tile, waves = 4 row groups x 2 channel halves;
// 64: 32x32 px tile, waves = 8 row groups.  Every wave owns 4 pixel rows (4 MFMA row tiles of 32 px) x 64 channels.
template <int NT>
struct PipeCfg {
    static constexpr int TH = NT == 128 ? 16 : 32;
    static constexpr int TW = 32, HWD = TW + 2, NP = (TH + 2) * HWD;  // 612 / 1156 halo pixels
    static constexpr int H_PIECES = (NP + 31) / 32;                    // 20 / 37 DMA pieces (1 KiB = 32 rows) per halo chunk
    static constexpr int H_STRIDE = H_PIECES * 1024;
    static constexpr int W_PIECES = 9 * NT / 32;                       // 36 / 18
    static constexpr int W_STRIDE = W_PIECES * 1024;
    static constexpr int HS = 3, WS = 2;                               // ring depths
    static constexpr int W_BASE = HS * H_STRIDE;
    static constexpr int BIAS_BASE = W_BASE + WS * W_STRIDE;           // [<= 512] floats of bias
    static constexpr int DUMP_BASE = BIAS_BASE + 2048;                 // 1 KiB: destination of the DMA pieces a wave does not have
    static constexpr int SMEM = DUMP_BASE + 1024;                      // 138240 / 153600 bytes: one block per CU
    static constexpr int HP_MAX = (H_PIECES + 7) / 8;                  // halo pieces per wave and chunk: 3 / 5 (some waves: one of them a dummy)
    static constexpr int WP_MAX = (W_PIECES + 7) / 8;                  // 5 / 3
    static constexpr int RG = TH / 4;                                  // row groups: 4 / 8
};

// one LDS-DMA piece: 64 lanes x 16 bytes -> LDS [lds_off, lds_off + 1 KiB) (wave-uniform), source = buffer base +
// per-lane voff + wave-uniform soff; lanes whose voff is out of range write zeros
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* smem, int lds_off, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(smem + lds_off), 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS operand reads as inline asm: hipcc sinks every compiler-visible ds_read to just before its MFMA and waits
// lgkmcnt(0) there (no software pipelining across taps); the asm forms pin the issue point, and the wait statement
// names every destination "+v" so that no consumer (and no register copy) is scheduled above it.
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8& dst, int addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_wait(bf16x8 (&av)[4], bf16x8 (&bv)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0]), "+v"(bv[1]));
}
// a_base = byte offset of halo row (4*rg)*HWD + lane%32 in the current stage; swz bit r*3+kw = swizzled half of the
// operand row (r, kw) for this lane (stage strides are multiples of 16 rows, so the bits do not depend on the stage).
// The four operand addresses of a tap are computed one tap ahead, among the MFMAs (the empty asm keeps hipcc from
// hoisting all 18 of them out of the loop into registers, and anchors the arithmetic where it is written).
template <int HWD, int TAP>
__device__ __forceinline__ void tap_addr(int (&addr)[4], int a_base, unsigned& swz) {
    constexpr int kh = TAP / 3, kw = TAP % 3;
    asm volatile("" : "+v"(swz));
#pragma unroll
    for (int i = 0; i < 4; ++i) addr[i] = a_base + ((i + kh) * HWD + kw) * 32 + (int)(((swz >> ((i + kh) * 3 + kw)) & 1u) << 4);
}
template <int NT, int TAP>
__device__ __forceinline__ void load_tap(bf16x8 (&av)[4], bf16x8 (&bv)[2], const int (&addr)[4], int b_addr) {
    lds_read16<(TAP * NT) * 32>(bv[0], b_addr);
    lds_read16<(TAP * NT + 32) * 32>(bv[1], b_addr);
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_read16<0>(av[i], addr[i]);
}

// MFMAs as inline asm too: the builtin is a pure value operation that instruction selection may place anywhere between
// its operands' definitions and its result's use — hipcc sank MFMAs across two and three taps, keeping their operands
// alive (36-42 spilled registers inside the loop).  asm volatile statements keep their program order.
// Hazards the compiler cannot see: a VALU read of an accumulator needs 12 wait states after the MFMA that wrote it
// (the epilogue is preceded by explicit s_nops); accumulate chains (D as the next C) need none.
__device__ __forceinline__ void mfma1(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma8(f32x16 (&acc)[4][2], const bf16x8 (&av)[4], const bf16x8 (&bv)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma1(acc[i][j], bv[j], av[i]);  // D^T = W x X^T: rows = channels
}

// q = n / d for n * d < 2^32 with the host-computed M = ceil(2^32 / d) (tile indices: a few thousand)
__device__ __forceinline__ int fastdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }  // magic 0: d = 1


// EPI: 0 = bias / activation / mask-out epilogue, 1 = data gradient gated by an activation tensor, 2 = by a bit mask
template <int NT, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_pipe_kernel(ConvKParams p, PipeArgs pa) {
    using Cfg = PipeCfg<NT>;
    constexpr int TH = Cfg::TH, HWD = Cfg::HWD, NP = Cfg::NP, H_PIECES = Cfg::H_PIECES, H_STRIDE = Cfg::H_STRIDE;
    constexpr int W_PIECES = Cfg::W_PIECES, W_STRIDE = Cfg::W_STRIDE, W_BASE = Cfg::W_BASE;
    constexpr int HP_MAX = Cfg::HP_MAX, WP_MAX = Cfg::WP_MAX, RG = Cfg::RG;
    static_assert(WP_MAX + HP_MAX <= 9, "one DMA piece per tap");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Ho, W = p.Wo, C = p.Ck, N = p.N;
    const int n_tiles = N / NT;
    const int tiles_x = (W + 31) >> 5, tiles_y = (H + TH - 1) / TH;
    const int tpi = tiles_x * tiles_y;

    // ---- static tile list: XCD x (blocks x, x+8, ...) owns a contiguous range, its blocks interleave inside it, so the
    // 32 CUs of an XCD work on 32 consecutive tiles (channel tiles of one pixel tile, then the neighbouring pixel tiles)
    const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int tq = pa.total_tiles >> 3, tr = pa.total_tiles & 7;
    const int xs = xcd * tq + (xcd < tr ? xcd : tr), xn = tq + (xcd < tr ? 1 : 0);
    if (bslot >= xn) return;
    // de-phase the CUs (experiment, dbg >> 8 = units of ~0.5 us): all blocks start together and every tile takes the same
    // time, so without it all 256 CUs issue their 128 KiB of output stores in the same microsecond
    for (int i = (bslot & 3) * (PIPE_DBG >> 8); i > 0; --i) __builtin_amdgcn_s_sleep(16);
    const int my_tiles = (xn - bslot + nslots - 1) / nslots;
    const int nchunks = C >> 4;
    const int total = my_tiles * nchunks;

    auto decode = [&](int k, int& b, int& y0, int& x0, int& n0) {
        const int t = xs + bslot + k * nslots;
        int pt = fastdiv(t, pa.m_ntiles);
        n0 = (t - pt * n_tiles) * NT;
        b = fastdiv(pt, pa.m_tpi);
        pt -= b * tpi;
        const int ty = fastdiv(pt, pa.m_tx);
        y0 = ty * TH;
        x0 = (pt - ty * tiles_x) * 32;
    };

    const unsigned bytes_x = (unsigned)((long)p.B * H * W * C * 2);
    const unsigned bytes_w = (unsigned)((long)N * 9 * C * 2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)bytes_x, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)bytes_w, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;  // beyond every tensor this kernel accepts: the bounds check returns zeros

    // ---- per-lane staging constants.  A DMA piece writes LDS rows piece*32 + (lane >> 1), physical half lane & 1.
    // Every wave issues exactly HP_MAX halo + WP_MAX weight pieces per chunk, without a branch: a piece that does not
    // exist for this wave (20 = 4x3 + 4x2 halo pieces ...) reads out of range and lands in a 1 KiB dump row.
    const int lr = lane >> 1, pslot = lane & 1;
    const int half = pslot ^ ((lr >> 3) & 1);  // logical 8-channel half this lane fetches (bank swizzle by bit 3 of the row)
    int h_dst[HP_MAX], w_dst[WP_MAX];  // LDS byte offset of the piece inside its stage, or -1 (dump)
#pragma unroll
    for (int it = 0; it < HP_MAX; ++it) h_dst[it] = wave + 8 * it < H_PIECES ? (wave + 8 * it) * 1024 : -1;
    unsigned voff_h[HP_MAX];
    auto halo_voff = [&](int b, int y0, int x0) {
#pragma unroll
        for (int it = 0; it < HP_MAX; ++it) {
            const int piece = wave + 8 * it;
            const int hp = piece * 32 + lr;
            const int hh = hp / HWD, ww = hp - hh * HWD;
            const int y = y0 - 1 + hh, x = x0 - 1 + ww;
            const bool ok = piece < H_PIECES && hp < NP && y >= 0 && y < H && x >= 0 && x < W;
            voff_h[it] = ok ? ((unsigned)((b * H + y) * W + x) * (unsigned)C + (unsigned)half * 8u) * 2u : OOB;
        }
    };
    unsigned voff_w[WP_MAX];
#pragma unroll
    for (int it = 0; it < WP_MAX; ++it) {
        const int piece = wave + 8 * it;
        const int r = piece * 32 + lr;  // weight row = tap * NT + n
        const int tap = r / NT, n = r - tap * NT;
        const int gt = p.flip_taps ? 8 - tap : tap;
        voff_w[it] = piece < W_PIECES ? ((unsigned)(n * 9 + gt) * (unsigned)C + (unsigned)half * 8u) * 2u : OOB;
        w_dst[it] = piece < W_PIECES ? piece * 1024 : -1;
    }
    // (plain macros, not lambdas: a lambda that captures a buffer resource makes the HOST pass drop the kernel's stub
    // without a diagnostic)
#define issue_h1(IT, STAGE, SOFF, MORE)                                                                              \
    dma16(rx, smem, h_dst[IT] >= 0 ? (STAGE) * H_STRIDE + h_dst[IT] : Cfg::DUMP_BASE, (MORE) ? voff_h[IT] : OOB, SOFF)
#define issue_w1(IT, WBUF, SOFF, MORE)                                                                               \
    dma16(rw, smem, w_dst[IT] >= 0 ? W_BASE + (WBUF) * W_STRIDE + w_dst[IT] : Cfg::DUMP_BASE, (MORE) ? voff_w[IT] : OOB, SOFF)

    // ---- operand addressing.  wave = (row group rg, channel half nh); MFMA row tile i = pixel row 4*rg + i.
    const int rg = wave % RG, nh = wave / RG;
    const int li = lane & 31, lk = lane >> 5;
    const int a_row = (4 * rg) * HWD + li;
    int a_base = a_row * 32;
    unsigned swz = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int t = a_row + r * HWD + kw;
            swz |= (unsigned)(((t >> 3) ^ lk) & 1) << (r * 3 + kw);
        }
    // weight rows tap*NT + nh*64 + j*32 + li: bit3(row) = bit3(li)
    const int b_cur0 = W_BASE + (nh * 64 + li) * 32 + ((((li >> 3) ^ lk) & 1) << 4), b_cur1 = b_cur0 + W_STRIDE;

    // bias -> LDS (read back with ds_read in the epilogue: no VMEM load there).  Staged before the first DMA is issued:
    // its ds_write retires behind the first lgkmcnt(0) of every wave, and every wave passes a barrier before an epilogue.
    if (EPI == 0 && (p.flags & STYLEX_EPI_BIAS) && tid < N) reinterpret_cast<float*>(smem + Cfg::BIAS_BASE)[tid] = p.bias[tid];

    f32x16 acc[4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    // ---- epilogue of one tile, straight from the accumulators (as conv_halo_dma.hip: D^T = W x X^T, a lane holds 4
    // consecutive channels of one pixel per quad; v_permlane32_swap pairs the half-waves' quads into 16-byte stores)
    const bool act = EPI == 0 && (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) != 0;
    const float slope = (p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f;
    const float gslope = p.res_scale;
    const bool mask_out = EPI == 0 && p.mask != nullptr;
    auto gatem = [&](unsigned u, unsigned bits) -> unsigned {
        const float a0 = __uint_as_float(u << 16), c0 = __uint_as_float(u & 0xffff0000u);
        f32x2_t t = {(bits & 1u) ? a0 : gslope * a0, (bits & 2u) ? c0 : gslope * c0};
        bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
        return *reinterpret_cast<unsigned*>(&r);
    };
    auto gate2 = [&](unsigned u, unsigned g) -> unsigned {
        const float a0 = __uint_as_float(u << 16), c0 = __uint_as_float(u & 0xffff0000u);
        const float ga = __uint_as_float(g << 16), gc = __uint_as_float(g & 0xffff0000u);
        f32x2_t t = {ga > 0.f ? a0 : gslope * a0, gc > 0.f ? c0 : gslope * c0};
        bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
        return *reinterpret_cast<unsigned*>(&r);
    };
    // Addressing: buffer stores / loads through descriptors of y (and the gate tensor / bit masks, which share its
    // geometry).  soffset = the tile's first output element + image row i (uniform, SALU), voffset = what a lane adds
    // to it — pixel lj of row 4*rg, channel half nh, quad half lh: 32 bits, the same for every tile — and (j, q) are
    // instruction immediates.  Pixels outside the image get an out-of-range voffset: the store is dropped (a gate load
    // returns 0) by the bounds check, so there is no exec-mask branch and no 64-bit vector arithmetic per store (the
    // first version spent more issue slots on v_mad_u64 / v_lshl_add_u64 / s_and_saveexec than on the stores).
    const unsigned bytes_y = (unsigned)((long)p.B * H * W * N * 2);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)bytes_y, 0x00020000);
    const __amdgpu_buffer_rsrc_t rgate = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, (int)bytes_y, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        EPI == 2 ? const_cast<unsigned char*>(p.gate_mask) : p.mask, 0, (int)(bytes_y >> 4), 0x00020000);
    const unsigned rowb = (unsigned)(W * N) * 2u;  // bytes between image rows
    auto pack2 = [](float a, float c) -> unsigned {  // one v_cvt_pk_bf16_f32 (RNE), low half = a
        f32x2_t t = {a, c};
        bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
        return *reinterpret_cast<unsigned*>(&r);
    };
    // FULL-LINE STORES.  After the permlane32 pairing a lane holds, for pixel p = lane % 32 of a row tile, the four
    // 16-byte pieces S = 4j + 2q + lh of the wave's 128-byte channel run — a store instruction per (j, q) would touch 32
    // different 128-byte lines with 32 bytes each.  Measured (ablation dbg 16): the same bytes written as whole lines
    // take 17 % off 64->64 @256^2 and 8-14 % off the 128^2 layers — the partial-line write requests, not the bytes,
    // were what the stores cost.  So the register index m = 2j + q is transposed with lane bits 4:3 (pixel / 8) first:
    // bit 3 <-> q by a DPP row rotate by 8 with a bank mask (one instruction per dword), bit 4 <-> j by
    // v_permlane16_swap.  Afterwards register k holds pixels 8k .. 8k+7, lane = (lh, j, q, pixel % 8): one instruction
    // stores eight complete lines.  The coalescing is by ADDRESS, so the lanes need no further reordering.
    const unsigned lane_off = (unsigned)(((4 * rg) * W + (lane & 7)) * N + nh * 64) * 2u +
                              (unsigned)(4 * ((lane >> 4) & 1) + 2 * ((lane >> 3) & 1) + (lane >> 5)) * 16u;
    const unsigned pixb8 = (unsigned)N * 16u;  // bytes between pixel p and p + 8
    auto epilogue = [&](int b, int y0, int x0, int n0) {
        const int lh = lk;
        const int nb = n0 + nh * 64;
        const unsigned t_off = __builtin_amdgcn_readfirstlane((unsigned)((((long)(b * H + y0) * W + x0) * N + n0) * 2));
        // operand-side addressing (gates, bit masks): pixel li, pieces (j, q, lh) — the layout before the transpose
        const unsigned g_off = x0 + li < W ? (unsigned)(((4 * rg) * W + li) * N + nh * 64 + 8 * lh) * 2u : OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool row_ok = y0 + 4 * rg + i < H;
            const unsigned soff = t_off + i * rowb;
            u32x4 R[4];  // R[2j + q]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x4 gv[2];
                unsigned gm[2];
                if (EPI != 0) {
                    const unsigned voff = row_ok ? g_off : OOB;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (EPI == 1) gv[q] = __builtin_amdgcn_raw_buffer_load_b128(rgate, voff + (j * 64 + q * 32), soff, 0);
                        if (EPI == 2) gm[q] = __builtin_amdgcn_raw_buffer_load_b8(rmask, (voff >> 4) + (j * 4 + q * 2), soff >> 4, 0);
                    }
                }
                float4 b4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    b4[g] = (EPI == 0 && (p.flags & STYLEX_EPI_BIAS))
                                ? *reinterpret_cast<const float4*>(smem + Cfg::BIAS_BASE + (nb + j * 32 + 8 * g + 4 * lh) * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned P[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[i][j][4 * g + 0] + b4[g].x, v1 = acc[i][j][4 * g + 1] + b4[g].y;
                    float v2 = acc[i][j][4 * g + 2] + b4[g].z, v3 = acc[i][j][4 * g + 3] + b4[g].w;
                    if (act) {
                        v0 = v0 > 0.f ? v0 : slope * v0;
                        v1 = v1 > 0.f ? v1 : slope * v1;
                        v2 = v2 > 0.f ? v2 : slope * v2;
                        v3 = v3 > 0.f ? v3 : slope * v3;
                    }
                    P[g][0] = pack2(v0, v1);
                    P[g][1] = pack2(v2, v3);
                }
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
                    uint4 v = make_uint4(P[2 * q][0], P[2 * q][1], P[2 * q + 1][0], P[2 * q + 1][1]);
                    if (EPI == 1) {
                        v.x = gate2(v.x, gv[q].x);
                        v.y = gate2(v.y, gv[q].y);
                        v.z = gate2(v.z, gv[q].z);
                        v.w = gate2(v.w, gv[q].w);
                    } else if (EPI == 2) {
                        const unsigned m = gm[q];
                        v.x = gatem(v.x, m);
                        v.y = gatem(v.y, m >> 2);
                        v.z = gatem(v.z, m >> 4);
                        v.w = gatem(v.w, m >> 6);
                    }
                    if (mask_out && !(PIPE_DBG & 1))  // one byte per 8 channels, addressed in the pre-transpose layout
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)stylex_sign_bits8(v), rmask,
                                                             ((row_ok ? g_off : OOB) >> 4) + (j * 4 + q * 2), soff >> 4, 0);
                    R[2 * j + q] = u32x4{v.x, v.y, v.z, v.w};
                }
            }
            // transpose register index (j, q) <-> lane bits (4, 3)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int d = 0; d < 4; ++d) {  // q <-> lane bit 3: lanes 8-15 of a row take the partner's q = 1 ... register
                    const unsigned a0 = R[2 * j][d], a1 = R[2 * j + 1][d];
                    R[2 * j][d] = (unsigned)__builtin_amdgcn_update_dpp((int)a0, (int)a1, 0x128, 0xF, 0xC, false);
                    R[2 * j + 1][d] = (unsigned)__builtin_amdgcn_update_dpp((int)a1, (int)a0, 0x128, 0xF, 0x3, false);
                }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int d = 0; d < 4; ++d) {  // j <-> lane bit 4: odd 16-lane rows of R[q] <-> even rows of R[2 + q]
                    auto r = __builtin_amdgcn_permlane16_swap(R[q][d], R[2 + q][d], false, false);
                    R[q][d] = r[0];
                    R[2 + q][d] = r[1];
                }
            if (!(PIPE_DBG & 1)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {  // register k = 2 * (pixel bit 4) + (pixel bit 3): pixels 8k .. 8k + 7
                    const unsigned voff = (row_ok && x0 + 8 * k + (lane & 7) < W) ? lane_off : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(R[k], ry, voff, soff + k * pixb8, PIPE_STORE_AUX);
                    // A VMEM store of more than 8 bytes needs a wait state before a VALU write of its data registers.
                    // hipcc does not pad it for a buffer store with an SGPR soffset (LLVM's rule ties the hazard to an
                    // immediate soffset), yet gfx950 showed it: `buffer_store_dwordx4 v[172:175]` directly followed by
                    // `v_and_b32 v172, ...` stored the NEW v172 in lanes 12-15 / 28-31 of every row of 16.  The data
                    // registers are operands of the nop statement, so nothing overwrites them before it, and the
                    // memory clobber keeps the statement behind the store.
                    asm volatile("s_nop 1" : "+v"(R[k]) : : "memory");
                }
            }
        }
    };

    // ---- producer state (uniform): the halo stream runs two chunks ahead of the consumer, the weight stream one
    int hb, hy0, hx0, hn0;                 // tile of the halo stream
    int h_k = 0, h_ch = 0, h_stage = 0;
    int w_k = 0, w_ch = 0, w_buf = 0, w_n0;
    int cb, cy0, cx0, cn0;                 // consumer tile
    decode(0, cb, cy0, cx0, cn0);
    hb = cb, hy0 = cy0, hx0 = cx0, hn0 = cn0;
    w_n0 = cn0;
    halo_voff(hb, hy0, hx0);
    auto adv_h = [&]() {
        h_stage = h_stage == 2 ? 0 : h_stage + 1;
        if (++h_ch == nchunks) {
            h_ch = 0;
            if (++h_k < my_tiles) {
                decode(h_k, hb, hy0, hx0, hn0);
                if (PIPE_DBG & 2) hb = 0, hy0 = 0, hx0 = 0;
                halo_voff(hb, hy0, hx0);
            }
        }
    };
    auto adv_w = [&]() {
        w_buf ^= 1;
        if (++w_ch == nchunks) {
            w_ch = 0;
            if (++w_k < my_tiles) {
                int b_, y_, x_;
                decode(w_k, b_, y_, x_, w_n0);
                if (PIPE_DBG & 4) w_n0 = 0;
            }
        }
    };
    // prologue: H(0), W(0), H(1) — the issue order the counted waits below rely on
#pragma unroll
    for (int it = 0; it < HP_MAX; ++it) issue_h1(it, 0, 0u, true);
    adv_h();
#pragma unroll
    for (int it = 0; it < WP_MAX; ++it) issue_w1(it, 0, (unsigned)(w_n0 * 9 * C) * 2u, true);
    adv_w();
#pragma unroll
    for (int it = 0; it < HP_MAX; ++it) issue_h1(it, 1, (unsigned)h_ch * 32u, total > 1);
    if (total > 1) adv_h();

    // ---- main loop, two chunks per iteration (nchunks is even; the operand register sets A / B alternate per tap and
    // nine taps is odd).  Per tap: wait for this tap's operands (requested one tap ago), request the next tap's, eight
    // MFMAs, DMA issue.  The barrier that publishes chunk g+1 sits in front of tap 8's MFMAs of chunk g — by then every
    // LDS read of chunk g has returned — so the first operands of chunk g+1 are in flight under those MFMAs.
    int c_ch = 0, c_k = 0, c_stage = 0;
    bf16x8 avA[4], bvA[2], avB[4], bvB[2];
    int addr[4];
    tap_addr<HWD, 0>(addr, a_base, swz);
    wait_vmcnt<HP_MAX>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_tap<NT, 0>(avA, bvA, addr, b_cur0);
    tap_addr<HWD, 1>(addr, a_base, swz);
    auto next_stage = [&]() {
        a_base += c_stage == 2 ? -2 * H_STRIDE : H_STRIDE;
        c_stage = c_stage == 2 ? 0 : c_stage + 1;
    };
    // DMA of one chunk spread over its first taps, PPT pieces per tap: weights of chunk g+1 first, then the halo of g+2
    constexpr int PPT = 2;
#define PIPE_DMA(TAP, MORE_W, MORE_H)                                                              \
    _Pragma("unroll") for (int q_ = (TAP) * PPT; q_ < (TAP) * PPT + PPT; ++q_) {                   \
        if (q_ < WP_MAX) issue_w1(q_, w_buf, w_soff, MORE_W);                                      \
        else if (q_ - WP_MAX < HP_MAX) issue_h1(q_ - WP_MAX, h_stage, h_soff, MORE_H);             \
    }
#define PIPE_TAP(TAP, AVC, BVC, AVN, BVN, BCUR, MORE_W, MORE_H) \
    lds_wait(AVC, BVC);                                        \
    load_tap<NT, TAP + 1>(AVN, BVN, addr, BCUR);               \
    __builtin_amdgcn_sched_barrier(0);                         \
    mfma8(acc, AVC, BVC);                                      \
    if (TAP == 7) next_stage();                                \
    tap_addr<HWD, (TAP + 2) % 9>(addr, a_base, swz);           \
    PIPE_DMA(TAP, MORE_W, MORE_H)                              \
    __builtin_amdgcn_sched_barrier(0);
#define PIPE_LAST(AVC, BVC, AVN, BVN, BNEXT)                   \
    lds_wait(AVC, BVC);                                        \
    if (PIPE_DBG & 8) wait_vmcnt<HP_MAX + 16>();                 \
    else wait_vmcnt<HP_MAX>();                                 \
    __builtin_amdgcn_s_barrier();                              \
    asm volatile("" ::: "memory");                             \
    load_tap<NT, 0>(AVN, BVN, addr, BNEXT);                    \
    __builtin_amdgcn_sched_barrier(0);                         \
    mfma8(acc, AVC, BVC);                                      \
    tap_addr<HWD, 1>(addr, a_base, swz);                       \
    __builtin_amdgcn_sched_barrier(0);
#define PIPE_CHUNK(A1, B1, A2, B2, BCUR, BNEXT, MORE_W, MORE_H)                                    \
    {                                                                                              \
        const bool mw_ = MORE_W, mh_ = MORE_H;                                                     \
        const unsigned w_soff = (unsigned)(w_n0 * 9 * C + w_ch * 16) * 2u, h_soff = (unsigned)h_ch * 32u; \
        PIPE_TAP(0, A1, B1, A2, B2, BCUR, mw_, mh_)                                                \
        PIPE_TAP(1, A2, B2, A1, B1, BCUR, mw_, mh_)                                                \
        PIPE_TAP(2, A1, B1, A2, B2, BCUR, mw_, mh_)                                                \
        PIPE_TAP(3, A2, B2, A1, B1, BCUR, mw_, mh_)                                                \
        PIPE_TAP(4, A1, B1, A2, B2, BCUR, mw_, mh_)                                                \
        PIPE_TAP(5, A2, B2, A1, B1, BCUR, mw_, mh_)                                                \
        PIPE_TAP(6, A1, B1, A2, B2, BCUR, mw_, mh_)                                                \
        PIPE_TAP(7, A2, B2, A1, B1, BCUR, mw_, mh_)                                                \
        PIPE_LAST(A1, B1, A2, B2, BNEXT)                                                           \
        if (mw_) adv_w();                                                                          \
        if (mh_) adv_h();                                                                          \
    }
    for (int g = 0; g < total; g += 2) {
        PIPE_CHUNK(avA, bvA, avB, bvB, b_cur0, b_cur1, g + 1 < total, g + 2 < total)
        PIPE_CHUNK(avB, bvB, avA, bvA, b_cur1, b_cur0, g + 2 < total, g + 3 < total)
        c_ch += 2;
        if (c_ch == nchunks) {  // tile complete: its stores drain under the next tile's MFMAs
            c_ch = 0;
            // MFMA results -> VALU readers (see mfma1).  The accumulators are operands of the statement: a plain
            // "memory" clobber does not stop hipcc from scheduling the epilogue's first accumulator reads ABOVE it
            // (seen: the lanes / rows the last MFMA passes write came out as garbage in one epilogue variant).
            asm volatile("s_nop 15\n\ts_nop 15"
                         : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),
                           "+v"(acc[3][0]), "+v"(acc[3][1]));
            epilogue(cb, cy0, cx0, cn0);
            if (++c_k < my_tiles) {
                decode(c_k, cb, cy0, cx0, cn0);
                zero_acc();
            }
        }
    }
#undef PIPE_CHUNK
#undef PIPE_LAST
#undef PIPE_TAP
#undef PIPE_DMA
#undef issue_h1
#undef issue_w1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the look-ahead operand reads of the (non-existent) next chunk
    wait_vmcnt<0>();  // dump-row DMAs of the stream's tail must not outlive the block's LDS allocation
}

int g_num_cus = 0;

unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

template <int NT, int EPI>
int launch_pipe(const ConvKParams& p, hipStream_t s) {
    using Cfg = PipeCfg<NT>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_pipe_kernel<NT, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (!g_num_cus) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        g_num_cus = n > 0 ? (n & ~7) : 256;
        if (g_num_cus < 8) g_num_cus = 8;
    }
    const int tiles_x = (p.Wo + 31) / 32, tiles_y = (p.Ho + Cfg::TH - 1) / Cfg::TH, n_tiles = p.N / NT;
    PipeArgs pa;
    pa.total_tiles = p.B * tiles_x * tiles_y * n_tiles;
    pa.m_ntiles = magic_of(n_tiles);
    pa.m_tpi = magic_of(tiles_x * tiles_y);
    pa.m_tx = magic_of(tiles_x);
#ifdef STYLEX_PIPE_ABLATION
    pa.dbg = getenv("STYLEX_PIPE_DBG") ? atoi(getenv("STYLEX_PIPE_DBG")) : 0;
#else
    pa.dbg = 0;
#endif
    stylex_note_kernel("conv3x3_pipe_kernel<%d, %d>", NT, EPI);
    hipLaunchKernelGGL((conv3x3_pipe_kernel<NT, EPI>), dim3((unsigned)g_num_cus), dim3(512), Cfg::SMEM, s, p, pa);
    return (int)hipGetLastError();
}

template <int NT>
int launch_pipe_epi(const ConvKParams& p, hipStream_t s) {
    if (p.flags & STYLEX_EPI_GATE) return launch_pipe<NT, 1>(p, s);
    if (p.flags & STYLEX_EPI_GATE_MASK) return launch_pipe<NT, 2>(p, s);
    return launch_pipe<NT, 0>(p, s);
}

}  // namespace

// STYLEX_NOT_APPLICABLE unless: bf16 activations, plain 3x3/s1/p1 without per-sample scales / noise / residual /
// space-to-depth, whole 16-channel chunks (>= 64 input channels), an output width that is a multiple of 64, at
// least one tile's worth of image, tensors below 2 GiB (32-bit buffer offsets).
int stylex_launch_pipe(const ConvKParams& p, hipStream_t s) {
    const char* env = getenv("STYLEX_CONV_PIPE");  // read per launch: A/B tests toggle it in-process
    if (env && env[0] == '0') return STYLEX_NOT_APPLICABLE;
    if (!p.act_bf16 || p.a_scale || p.s2d_c) return STYLEX_NOT_APPLICABLE;
    if (p.flags & ~(STYLEX_EPI_BIAS | STYLEX_EPI_LRELU | STYLEX_EPI_RELU | STYLEX_EPI_GATE | STYLEX_EPI_GATE_MASK | STYLEX_EPI_MASK_OUT))
        return STYLEX_NOT_APPLICABLE;
    if ((p.flags & STYLEX_EPI_GATE) && (!p.residual || (reinterpret_cast<uintptr_t>(p.residual) & 15))) return STYLEX_NOT_APPLICABLE;
    if (p.Ck < 64 || p.Ck % 32 != 0 || p.N % 64 != 0 || p.N > 512 || p.Wo < 32 || p.Ho < 16) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (reinterpret_cast<uintptr_t>(p.y) & 15))
        return STYLEX_NOT_APPLICABLE;
    if ((long)p.B * p.Ho * p.Wo * p.Ck * 2 >= (1l << 31) || (long)p.N * 9 * p.Ck * 2 >= (1l << 31) ||
        (long)p.B * p.Ho * p.Wo * p.N * 2 >= (1l << 31))
        return STYLEX_NOT_APPLICABLE;
    // every eligibility decision comes BEFORE the dry-run return, so that stylex_conv_mask_supported() (a dry launch)
    // reports exactly what a real launch would do
    const bool n128 = p.N % 128 == 0 && !(env && env[0] == '6');
    if (!n128) {
        if (p.Ho < 32 && !(env && env[0] == '6')) return STYLEX_NOT_APPLICABLE;  // 32x32 px tiles
        // measured (tools/bench_pipe.py, B = 128): on the 64-channel-output tiles the per-tile kernel keeps a 3-6 % edge
        // when the epilogue reads a whole gate TENSOR or stores nothing but the plain 64 -> 64 data gradient
        if (!(env && env[0] == '6') && ((p.flags & STYLEX_EPI_GATE) || (p.flip_taps && !p.flags && p.Ck == 64)))
            return STYLEX_NOT_APPLICABLE;
    }
    if (p.dry) return 0;
    if (n128) return launch_pipe_epi<128>(p, s);
    return launch_pipe_epi<64>(p, s);
}
