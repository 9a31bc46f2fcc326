// conv_line64.hip — 3x3 / stride-1 / pad-1 forward and data gradient of the 64 -> 64 channel layers at 128^2 / 256^2
// (conv2 of the first DiscriminatorBlock of D and of the encoder, /root/reference/stylex/stylex_train.py:724-736), bf16
// NHWC.  Round 4.
//
// Why a kernel of their own.  These launches move 1.07 GB for 77 GFLOP (B = 64): they are bound by the memory side,
// and the pipelined kernel (conv_pipe.hip) ran them at 2.8-3.1 TB/s with its matrix pipe 44 % busy.  Its K loop walks
// 16-channel chunks, so every LDS-DMA instruction fetches 32 bytes from each of 32 different 128-byte pixel lines
// (tools/line_granularity_probe.hip: 3-4.5 TB/s for that pattern against 5.5-6 for whole lines).  With 64 input
// channels a pixel IS one 128-byte line, so here
//   * the K stage is the whole pixel: a DMA instruction fetches 8 complete lines (8 pixels x 128 bytes), four times
//     fewer line look-ups per byte, and every input line is requested once per tile instead of four times;
//   * the whole weight tensor (9 taps x 64 x 64 bf16 = 72 KiB) is loaded into LDS ONCE per block and stays there — the
//     block is persistent and walks a static, XCD-contiguous tile list (as conv_pipe.hip);
//   * a tile is 8 rows x 32 pixels: its halo (10 x 34 pixels x 128 B = 42.5 KiB) is double-buffered, 72 + 2 x 43 KiB
//     of the CU's 160 KiB; wave w of the 8 owns pixel row w (32 px x 64 channels = 2 MFMA tiles, 72 MFMAs per tile);
//   * one barrier per tile; the halo of tile k+1 is in flight under the MFMAs and the stores of tile k, and the wait
//     for it is counted (`vmcnt(4)`: the previous tile's stores may still be in flight — vmcnt is one in-order queue).
// LDS rows are 128 bytes (8 slots of 16 B = 8 channels); slot s of row R lives at physical slot s ^ ((R >> 1) & 7): an
// MFMA operand read (16 lanes = 16 consecutive rows, one logical slot) then covers all 64 banks exactly once.
// Epilogue as conv_pipe.hip: from the accumulators, register transpose to whole-line stores, bias / LeakyReLU /
// mask-out (EPI 0) or the data gradient's gate as a bit mask (EPI 2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

// cache policy of the output stores (buffer_store aux bits; 2 = nt, streaming: tools/bench_s2d_dgrad.py A/B, DESIGN §3 "Round 5")
#ifndef LN_STORE_AUX
#define LN_STORE_AUX 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct StylexLineArgs {
    int total_tiles;
    unsigned m_tpi, m_tx;  // magic reciprocals of tiles per image, tiles_x
};

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int TH = 8, TW = 32, HWD = TW + 2, NP = (TH + 2) * HWD;  // 340 halo pixels
constexpr int H_PIECES = (NP + 7) / 8;                               // 43 DMA pieces of 8 pixels (1 KiB)
constexpr int H_STAGE = H_PIECES * 1024;
constexpr int W_PIECES = 9 * 64 / 8;                                 // 72
constexpr int W_BASE = 0, H_BASE = W_PIECES * 1024;
constexpr int DUMP_BASE = H_BASE + 2 * H_STAGE;                           // destination of the pieces a wave does not have
constexpr int SMEM = DUMP_BASE + 1024;                               // 162816 bytes: one block per CU
constexpr int HP = (H_PIECES + 7) / 8;                               // 6 halo pieces per wave and tile (waves 3-7: one dummy)
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* smem, int lds_off, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(smem + lds_off), 16, voff, 0, 0, 0);
}
__device__ __forceinline__ int fastdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

template <int EPI>
__global__ __launch_bounds__(512) void conv3x3_line64_kernel(ConvKParams p, StylexLineArgs la) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Ho, W = p.Wo;
    const int tiles_x = W >> 5, tiles_y = H >> 3, tpi = tiles_x * tiles_y;

    // static tile list, XCD-contiguous (as conv_pipe.hip)
    const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int tq = la.total_tiles >> 3, tr = la.total_tiles & 7;
    const int xs = xcd * tq + (xcd < tr ? xcd : tr), xn = tq + (xcd < tr ? 1 : 0);
    if (bslot >= xn) return;
    const int my_tiles = (xn - bslot + nslots - 1) / nslots;
    auto decode = [&](int k, int& b, int& y0, int& x0) {
        int pt = xs + bslot + k * nslots;
        b = fastdiv(pt, la.m_tpi);
        pt -= b * tpi;
        const int ty = fastdiv(pt, la.m_tx);
        y0 = ty * TH;
        x0 = (pt - ty * tiles_x) * TW;
    };

    const unsigned bytes_x = (unsigned)((long)p.B * H * W * 64 * 2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)bytes_x, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, 64 * 9 * 64 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)bytes_x, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        EPI == 2 ? const_cast<unsigned char*>(p.gate_mask) : p.mask, 0, (int)(bytes_x >> 4), 0x00020000);

    // ---- the weights, once: LDS row wr = tap * 64 + n, a DMA piece = 8 rows, lane -> (row piece * 8 + lane / 8, physical slot lane % 8)
    const int prow = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int it = 0; it < W_PIECES / 8; ++it) {
        const int piece = wave + 8 * it;
        const int wr = piece * 8 + prow;
        const int ls = pslot ^ ((wr >> 1) & 7);
        const int tap = wr >> 6, n = wr & 63;
        const int gt = p.flip_taps ? 8 - tap : tap;
        dma16(rw, smem, W_BASE + piece * 1024, (unsigned)(((n * 9 + gt) * 64 + ls * 8) * 2));
    }

    // ---- halo of one tile into a stage: 43 pieces of 8 pixels, lane -> (halo pixel piece * 8 + lane / 8, slot lane % 8)
    auto issue_halo = [&](int k, int stage) {
        int b, y0, x0;
        decode(k, b, y0, x0);
#pragma unroll
        for (int it = 0; it < HP; ++it) {
            const int piece = wave + 8 * it;
            const int hp = piece * 8 + prow;
            const int hh = (hp * 241) >> 13, ww = hp - hh * HWD;  // hp / 34 for hp < 344
            const int y = y0 - 1 + hh, x = x0 - 1 + ww;
            const bool ok = piece < H_PIECES && hp < NP && y >= 0 && y < H && x >= 0 && x < W;
            const int ls = pslot ^ ((hp >> 1) & 7);
            const unsigned voff = ok ? (unsigned)((((b * H + y) * W + x) * 64 + ls * 8) * 2) : OOB;
            dma16(rx, smem, piece < H_PIECES ? H_BASE + stage * H_STAGE + piece * 1024 : DUMP_BASE, voff);
        }
    };

    // ---- operand addressing: wave = pixel row of the tile; lane li = pixel (A) / channel (B) of a 32-wide MFMA tile
    const int li = lane & 31, lk = lane >> 5;
    const int fw = (li >> 1) & 7;  // swizzle of weight row tap*64 + j*32 + li (the tap / j terms do not reach bits 1-3)
    int b_off[4];                  // byte offset of this lane's 16-byte k-slice kc inside weight row li
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) b_off[kc] = W_BASE + li * 128 + (((kc * 2 + lk) ^ fw) << 4);

    // ---- epilogue constants (conv_pipe.hip's register epilogue for ONE row tile)
    const bool act = EPI == 0 && (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) != 0;
    const float slope = (p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f;
    const float gslope = p.res_scale;
    const bool mask_out = EPI == 0 && p.mask != nullptr;
    auto pack2 = [](float a, float c) -> unsigned {
        f32x2_t t = {a, c};
        bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
        return *reinterpret_cast<unsigned*>(&r);
    };
    auto gatem = [&](unsigned u, unsigned bits) -> unsigned {
        const float a0 = __uint_as_float(u << 16), c0 = __uint_as_float(u & 0xffff0000u);
        return pack2((bits & 1u) ? a0 : gslope * a0, (bits & 2u) ? c0 : gslope * c0);
    };
    // whole-line stores after the register transpose: register k holds pixels 8k .. 8k+7, lane = (lh, j, q, pixel % 8)
    const unsigned lane_off = (unsigned)((lane & 7) * 64) * 2u + (unsigned)(4 * ((lane >> 4) & 1) + 2 * ((lane >> 3) & 1) + (lane >> 5)) * 16u;
    const unsigned g_off = (unsigned)(li * 64 + 8 * lk) * 2u;  // gates / masks: pixel li, pieces (j, q, lh) — before the transpose
    // bias in registers for the whole kernel (an LDS read in the epilogue would make hipcc drain the DMA queue there: it
    // puts vmcnt(0) in front of every LDS read it can see behind a load-to-LDS)
    float4 b4r[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            b4r[j][g] = (EPI == 0 && (p.flags & STYLEX_EPI_BIAS)) ? *reinterpret_cast<const float4*>(p.bias + j * 32 + 8 * g + 4 * lk)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 acc[2];
    issue_halo(0, 0);
    for (int k = 0; k < my_tiles; ++k) {
        // halo(k) — and at k = 0 the weights — must have landed.  Younger than halo(k) in the queue: only the stores of
        // tile k-1 (>= 4 per wave), so vmcnt(4) proves it without waiting for those stores.
        if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        // everyone's pieces of halo(k) are visible; everyone is done reading the other stage.  A RAW barrier (round 5):
        // __syncthreads() carries an s_waitcnt vmcnt(0) of its own, i.e. it waited for the stores of tile k-1 after all
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int b, y0, x0;
        decode(k, b, y0, x0);
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((((b * H + y0 + wave) * W + x0) * 64) * 2));
        // the gate bits of tile k are requested BEFORE the next halo: waiting for them in the epilogue then leaves the
        // halo's six pieces in flight (vmcnt is in-order)
        unsigned gm[2][2];
        if (EPI == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) gm[j][q] = __builtin_amdgcn_raw_buffer_load_b8(rmask, (g_off >> 4) + (j * 4 + q * 2), soff >> 4, 0);
        }
        if (k + 1 < my_tiles) issue_halo(k + 1, (k + 1) & 1);
        const char* hbase = smem + H_BASE + (k & 1) * H_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            const int hrow = (wave + kh) * HWD + li + kw;
            const int f = (hrow >> 1) & 7;
            const char* arow = hbase + hrow * 128;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + (((kc * 2 + lk) ^ f) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(smem + b_off[kc] + (tap * 64 + j * 32) * 128);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[j], 0, 0, 0);  // D^T: rows = channels
                }
            }
        }
        // ---- epilogue of tile k: pixel row y0 + wave, 32 pixels x 64 channels from the accumulators
        u32x4 R[4];  // R[2j + q]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4(&b4)[4] = b4r[j];
            unsigned P[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = acc[j][4 * g + 0] + b4[g].x, v1 = acc[j][4 * g + 1] + b4[g].y;
                float v2 = acc[j][4 * g + 2] + b4[g].z, v3 = acc[j][4 * g + 3] + b4[g].w;
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
                if (EPI == 2) {
                    const unsigned m = gm[j][q];
                    v.x = gatem(v.x, m);
                    v.y = gatem(v.y, m >> 2);
                    v.z = gatem(v.z, m >> 4);
                    v.w = gatem(v.w, m >> 6);
                }
                if (mask_out)  // one byte per 8 channels, addressed in the pre-transpose layout
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)stylex_sign_bits8(v), rmask, (g_off >> 4) + (j * 4 + q * 2), soff >> 4, 0);
                R[2 * j + q] = u32x4{v.x, v.y, v.z, v.w};
            }
        }
        // transpose register index (j, q) <-> lane bits (4, 3)  (conv_pipe.hip)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned a0 = R[2 * j][d], a1 = R[2 * j + 1][d];
                R[2 * j][d] = (unsigned)__builtin_amdgcn_update_dpp((int)a0, (int)a1, 0x128, 0xF, 0xC, false);
                R[2 * j + 1][d] = (unsigned)__builtin_amdgcn_update_dpp((int)a1, (int)a0, 0x128, 0xF, 0x3, false);
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                auto r = __builtin_amdgcn_permlane16_swap(R[q][d], R[2 + q][d], false, false);
                R[q][d] = r[0];
                R[2 + q][d] = r[1];
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {  // register kk: pixels 8kk .. 8kk + 7
            __builtin_amdgcn_raw_buffer_store_b128(R[kk], ry, lane_off, soff + kk * (8 * 64 * 2), LN_STORE_AUX);
            asm volatile("s_nop 1" : "+v"(R[kk]) : : "memory");  // VMEM store data hazard (see conv_pipe.hip)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dump-row DMAs must not outlive the block's LDS allocation
}

int g_line_cus = 0;
unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

template <int EPI>
int launch_line(const ConvKParams& p, hipStream_t s) {
    // 0 = not asked yet, 1 = granted, -1 = refused (a device with less than 160 KiB of LDS): the pipelined and per-tile
    // kernels behind this one in stylex_launch_halo's dispatch then serve the launch
    static int attr_state = 0;
    if (attr_state == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_line64_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_state = e == hipSuccess ? 1 : -1;
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (attr_state < 0) return STYLEX_NOT_APPLICABLE;
    if (!g_line_cus) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        g_line_cus = n > 0 ? (n & ~7) : 256;
        if (g_line_cus < 8) g_line_cus = 8;
    }
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    StylexLineArgs la;
    la.total_tiles = p.B * tiles_x * tiles_y;
    la.m_tpi = magic_of(tiles_x * tiles_y);
    la.m_tx = magic_of(tiles_x);
    stylex_note_kernel("conv3x3_line64_kernel<%d>", EPI);
    hipLaunchKernelGGL((conv3x3_line64_kernel<EPI>), dim3((unsigned)g_line_cus), dim3(512), SMEM, s, p, la);
    return (int)hipGetLastError();
}

}  // namespace

// STYLEX_NOT_APPLICABLE unless: bf16 activations, plain 3x3/s1/p1, 64 -> 64 channels, no per-sample scales / noise /
// residual / space-to-depth / gate TENSOR, an image of whole 8 x 32 tiles with at least 128 x 128 pixels (below that the
// pipelined kernel's larger tile wins), tensors below 2 GiB.  STYLEX_CONV_LINE64=0 switches it off.
int stylex_launch_line64(const ConvKParams& p, hipStream_t s) {
    const char* env = getenv("STYLEX_CONV_LINE64");
    if (env && env[0] == '0') return STYLEX_NOT_APPLICABLE;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.Hi != p.Ho || p.Wi != p.Wo) return STYLEX_NOT_APPLICABLE;
    if (!p.act_bf16 || p.a_scale || p.out_scale || p.noise || p.s2d_c || p.Ck != 64 || p.N != 64) return STYLEX_NOT_APPLICABLE;
    if (p.flags & ~(STYLEX_EPI_BIAS | STYLEX_EPI_LRELU | STYLEX_EPI_RELU | STYLEX_EPI_GATE_MASK | STYLEX_EPI_MASK_OUT)) return STYLEX_NOT_APPLICABLE;
    if ((p.flags & STYLEX_EPI_GATE_MASK) && (p.flags & (STYLEX_EPI_BIAS | STYLEX_EPI_LRELU | STYLEX_EPI_RELU | STYLEX_EPI_MASK_OUT)))
        return STYLEX_NOT_APPLICABLE;
    if (p.Wo % TW != 0 || p.Ho % TH != 0 || (long)p.Ho * p.Wo < 128 * 128) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (reinterpret_cast<uintptr_t>(p.y) & 15))
        return STYLEX_NOT_APPLICABLE;
    if ((p.flags & STYLEX_EPI_BIAS) && (!p.bias || (reinterpret_cast<uintptr_t>(p.bias) & 15))) return STYLEX_NOT_APPLICABLE;
    if ((long)p.B * p.Ho * p.Wo * 64 * 2 >= (1l << 31)) return STYLEX_NOT_APPLICABLE;
    if (p.dry) return 0;
    if (p.flags & STYLEX_EPI_GATE_MASK) return launch_line<2>(p, s);
    return launch_line<0>(p, s);
}
