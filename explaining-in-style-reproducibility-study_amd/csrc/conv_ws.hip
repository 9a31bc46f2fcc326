// conv_ws.hip — weights-stationary, persistent, streaming 3x3 / stride-1 / pad-1 convolution (forward and data
// gradient) for the LOW-CHANNEL HIGH-RESOLUTION layers: C, N in {32, 64} at 128^2 .. 256^2 (bf16 NHWC activations).
//
// Why a third conv kernel.  Those layers (G5/G6 of the generator, D0 of the discriminator / encoder) carry ~300 FLOPs
// per activation byte — at or below the ridge of the machine (2.5 PF / 8 TB/s = 312): they are HBM-side.  The
// per-tile kernels (conv_halo.hip, conv_halo_dma.hip) reach 1.0 - 3.1 TB/s on them (bench.py per-layer roofline:
// 13 - 39 % of the HBM roof) because a tile is only 2 - 4 K-chunks long: the prologue load latency, the weight
// re-staging and the epilogue are most of a block's life, and two resident blocks per CU do not hide them.  Here
//   * one block per CU lives for the whole launch and walks a contiguous range of 16x32-pixel tiles;
//   * the weights of ALL chunks (<= 72 KiB) are staged ONCE per block (once per sample for a modulated conv: the
//     per-sample modulation s[b][c] — or the demodulation d[b][n] of the data gradient — is folded into the staged
//     weights, which is exactly the reference's w2 * (w1 + 1), stylex_train.py:650-651, but only ever in LDS);
//   * the input halo streams through a 3-slot LDS ring by LDS-DMA, two (tile, chunk) items ahead of the MFMAs and
//     ACROSS tile boundaries, so the memory queue never drains: the kernel runs at the HBM rate.
//
// LDS image (one block per CU, 151.5 KiB): weights [chunk][tap][n] as 32-byte rows (16 bf16) | 3 halo slots of 20
// KiB (18x34 halo pixels as 32-byte rows, layout / swizzle of conv_halo_dma.hip) | 16 KiB epilogue transpose scratch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

__device__ uint4 g_zero_page_ws[4];

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gl_void_ptr;

__device__ __forceinline__ unsigned short to_bf16(float v) {
    f32x2_t t = {v, 0.f};
    bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
    return (unsigned short)(*reinterpret_cast<unsigned*>(&r) & 0xffffu);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

constexpr int TW = 32, TH = 16, HWD = TW + 2, NP = (TH + 2) * HWD;  // 612 halo pixels
constexpr int ROW = 32;                                              // bytes per LDS row (16 bf16)
constexpr int HALO_PIECES = (NP + 31) / 32;                          // 20 DMA pieces of 32 rows (1 KiB each)
constexpr int HALO_BYTES = HALO_PIECES * 1024;                       // 20 KiB per ring slot
constexpr int RING = 3;
constexpr int HP_PER_WAVE = HALO_PIECES / 4;                         // 5 DMA instructions per wave and item

template <int TNJ>
struct WsCfg {
    static constexpr int BN = TNJ * 32;
    static constexpr int W_CHUNK = 9 * BN * ROW;          // weight bytes of one 16-channel chunk
    static constexpr int MAXCH = 4;                       // C <= 64
    static constexpr int W_BYTES = MAXCH * W_CHUNK;       // 73728 (TNJ = 2) / 36864 (TNJ = 1)
    static constexpr int RING_OFF = W_BYTES;
    static constexpr int SCR_OFF = RING_OFF + RING * HALO_BYTES;
    static constexpr int SCR_BYTES = 4 * 32 * BN * 2;     // per-wave 32 px x BN bf16 transpose scratch
    static constexpr int SMEM_BYTES = SCR_OFF + SCR_BYTES;
};

template <int TNJ>
__global__ __launch_bounds__(256, 1) void conv3x3_ws_kernel(ConvKParams p) {
    using Cfg = WsCfg<TNJ>;
    constexpr int BN = Cfg::BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Ho, W = p.Wo, C = p.Ck, N = p.N;
    const int tiles_x = W / TW, tiles_y = H / TH, tiles_img = tiles_x * tiles_y;
    const int total_tiles = p.B * tiles_img;
    const int nch = C >> 4;

    // contiguous tile range of this block (so that a block rarely changes sample)
    const int per = (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per;
    const int t_end = min(t_begin + per, total_tiles);
    if (t_begin >= t_end) return;

    const unsigned short* xs = reinterpret_cast<const unsigned short*>(p.a);
    const unsigned short* ws = reinterpret_cast<const unsigned short*>(p.w);
    const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero_page_ws);
    const int lr = lane >> 1, pslot = lane & 1;
    const int slot = pslot ^ ((lr >> 3) & 1);  // logical channel half this lane fetches (bank swizzle, see conv_halo_dma)

    // tile-independent geometry of this lane's halo pieces
    int hh[HP_PER_WAVE], ww[HP_PER_WAVE];
    bool hin[HP_PER_WAVE];
#pragma unroll
    for (int it = 0; it < HP_PER_WAVE; ++it) {
        const int hp = (wave + 4 * it) * 32 + lr;
        hh[it] = hp / HWD;
        ww[it] = hp - hh[it] * HWD;
        hin[it] = hp < NP;
    }

    // ---- weights: staged once per block (per sample when scaled) ----------------------------------------------
    // LDS row (chunk ch, tap, n) = 16 channels ch*16 .. +15 of packed weight [n][tap][C]; half h at physical half
    // h ^ bit3(row) like every other row of this layout.
    auto stage_weights = [&](int b) {
        const float* sc = p.a_scale ? p.a_scale + (long)b * C : nullptr;
        const int rows = nch * 9 * BN;  // 32-byte rows
        for (int s = tid; s < rows * 2; s += 256) {
            const int row = s >> 1, ph = s & 1;
            const int ch = row / (9 * BN), rr = row - ch * (9 * BN);
            const int tap = rr / BN, n = rr - tap * BN;
            const int half = ph ^ ((row >> 3) & 1);
            const int gt = p.flip_taps ? 8 - tap : tap;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int c0 = ch * 16 + half * 8;
            if (n < N) v = *reinterpret_cast<const uint4*>(ws + ((long)n * 9 + gt) * C + c0);
            if (sc) {
                const float4 s0 = *reinterpret_cast<const float4*>(sc + c0), s1 = *reinterpret_cast<const float4*>(sc + c0 + 4);
                v.x = (unsigned)to_bf16(bf_lo(v.x) * s0.x) | ((unsigned)to_bf16(bf_hi(v.x) * s0.y) << 16);
                v.y = (unsigned)to_bf16(bf_lo(v.y) * s0.z) | ((unsigned)to_bf16(bf_hi(v.y) * s0.w) << 16);
                v.z = (unsigned)to_bf16(bf_lo(v.z) * s1.x) | ((unsigned)to_bf16(bf_hi(v.z) * s1.y) << 16);
                v.w = (unsigned)to_bf16(bf_lo(v.w) * s1.z) | ((unsigned)to_bf16(bf_hi(v.w) * s1.w) << 16);
            }
            *reinterpret_cast<uint4*>(smem + row * ROW + ph * 16) = v;
        }
    };

    // ---- halo DMA of one (tile, chunk) item into ring slot rs ---------------------------------------------------
    auto issue_item = [&](int tile, int ch, int rs) {
        const int b = tile / tiles_img;
        const int pt = tile - b * tiles_img;
        const int y0 = (pt / tiles_x) * TH, x0 = (pt % tiles_x) * TW;
        char* base = smem + Cfg::RING_OFF + rs * HALO_BYTES;
#pragma unroll
        for (int it = 0; it < HP_PER_WAVE; ++it) {
            const int y = y0 - 1 + hh[it], x = x0 - 1 + ww[it];
            const bool ok = hin[it] && y >= 0 && y < H && x >= 0 && x < W;
            const unsigned short* g = xs + ((long)(b * H + y) * W + x) * C + ch * 16 + slot * 8;
            const unsigned short* src = ok ? g : zero;
            __builtin_amdgcn_global_load_lds((gl_void_ptr)src, (lds_void_ptr)(base + (wave + 4 * it) * 1024), 16, 0, 0);
        }
    };

    // operand addressing (identical to conv_halo_dma.hip): MFMA row-tile i of this wave = tile row 4*wave + i
    const int li = lane & 31, lk = lane >> 5;
    const int a_row = (4 * wave) * HWD + li;
    int a_off[6][3];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int t = a_row + r * HWD + kw;
            a_off[r][kw] = (t * ROW) | ((((t >> 3) ^ lk) & 1) << 4);
        }
    const int b_lane = li * ROW + ((((li >> 3) ^ lk) & 1) << 4);  // + chunk*W_CHUNK + (tap*BN + j*32)*ROW

    const int lj = lane & 31, lh = lane >> 5;
    const bool act = (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) != 0;
    const float slope = (p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f;
    unsigned short* yout = reinterpret_cast<unsigned short*>(p.y);
    const unsigned short* gate = (p.flags & STYLEX_EPI_GATE) ? reinterpret_cast<const unsigned short*>(p.residual) : nullptr;
    const float gslope = p.res_scale;
    float bias[TNJ];
#pragma unroll
    for (int j = 0; j < TNJ; ++j) {
        const int n = j * 32 + lj;
        bias[j] = ((p.flags & STYLEX_EPI_BIAS) && n < N) ? p.bias[n] : 0.f;
    }

    // ---- prologue: weights of the first sample, first two items ---------------------------------------------------
    int wb = t_begin / tiles_img;  // sample whose (scaled) weights are resident
    stage_weights(wb);
    const int items = (t_end - t_begin) * nch;
    issue_item(t_begin, 0, 0);
    if (items > 1) issue_item(t_begin + (1 / nch), 1 % nch, 1);

    f32x16 acc[4][TNJ];
    // epilogue of one tile: out_scale / bias / activation on the fp32 accumulators, transpose through the wave's bf16
    // scratch, 16-byte row stores (+ the activation gate of the layer below for a data gradient)
    auto epilogue = [&](int b, int y0, int x0) {
        char* scr = smem + Cfg::SCR_OFF + wave * (32 * BN * 2);
        float osc[TNJ];
#pragma unroll
        for (int j = 0; j < TNJ; ++j) {
            const int n = j * 32 + lj;
            osc[j] = ((p.flags & STYLEX_EPI_OSCALE) && n < N) ? p.out_scale[(long)b * N + n] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < TNJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[i][j][r] * osc[j] + bias[j];
                    if (act) v = v > 0.f ? v : slope * v;
                    *reinterpret_cast<unsigned short*>(scr + px * (BN * 2) + (j * 32 + lj) * 2) = to_bf16(v);
                }
            const int y = y0 + 4 * wave + i;
#pragma unroll
            for (int q2 = 0; q2 < BN / 16; ++q2) {
                const int id = lane + 64 * q2;  // 32 px x (BN / 8) slots of 8 channels
                const int px = id / (BN / 8), q = id % (BN / 8);
                const int x = x0 + px, n = q * 8;
                if (n < N) {
                    uint4 v = *reinterpret_cast<const uint4*>(scr + px * (BN * 2) + q * 16);
                    const long o = ((long)(b * H + y) * W + x) * N + n;
                    if (gate) {
                        const uint4 gv = *reinterpret_cast<const uint4*>(gate + o);
                        auto g2 = [&](unsigned u, unsigned g) -> unsigned {
                            const float a = bf_lo(u), c = bf_hi(u);
                            return (unsigned)to_bf16(bf_lo(g) > 0.f ? a : gslope * a) |
                                   ((unsigned)to_bf16(bf_hi(g) > 0.f ? c : gslope * c) << 16);
                        };
                        v.x = g2(v.x, gv.x);
                        v.y = g2(v.y, gv.y);
                        v.z = g2(v.z, gv.z);
                        v.w = g2(v.w, gv.w);
                    }
                    *reinterpret_cast<uint4*>(yout + o) = v;
                }
            }
        }
    };

    int k = 0;
    int pb = 0, py0 = 0, px0 = 0;  // tile whose accumulators await their epilogue
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / tiles_img;
        const int pt = tile - b * tiles_img;
        const int y0 = (pt / tiles_x) * TH, x0 = (pt % tiles_x) * TW;
        for (int ch = 0; ch < nch; ++ch, ++k) {
            // Item k must have landed.  Loads complete in order, so "at most the 5 DMA instructions of item k+1
            // outstanding" proves it (stores of an epilogue may complete out of order with loads: they can only make
            // this wait longer, never shorter).
            if (k + 1 < items) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HP_PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // all waves' pieces of item k are visible; everyone is done with slot (k+2) % RING
            if (k + 2 < items) {
                const int k2 = k + 2;
                issue_item(t_begin + k2 / nch, k2 % nch, k2 % RING);
            }
            if (ch == 0) {
                // the previous tile's epilogue runs HERE, behind the DMA issue of item k+2 and a full item away from
                // the next wait, so its stores drain under MFMAs instead of in front of a vmcnt
                if (tile != t_begin) epilogue(pb, py0, px0);
                if (b != wb) {  // next sample: re-stage the (scaled) weights; every wave is past its reads of the old ones
                    __syncthreads();
                    stage_weights(b);
                    wb = b;
                    __syncthreads();
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TNJ; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            }
            const char* hb = smem + Cfg::RING_OFF + (k % RING) * HALO_BYTES;
            const char* wbp = smem + ch * Cfg::W_CHUNK + b_lane;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                bf16x8 av[4], bv[TNJ];
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const bf16x8*>(hb + a_off[i + kh][kw]);
#pragma unroll
                for (int j = 0; j < TNJ; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(wbp + (tap * BN + j * 32) * ROW);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TNJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        pb = b;
        py0 = y0;
        px0 = x0;
    }
    epilogue(pb, py0, px0);
}

}  // namespace

// STYLEX_NOT_APPLICABLE unless: bf16 activations, 3x3/s1/p1 (checked by the caller), C in {32, 64}, N in {32, 64},
// whole 16x32 tiles, enough tiles to give every CU a range, no noise / residual epilogue.
int stylex_launch_ws(const ConvKParams& p, hipStream_t s) {
    // OPT-IN (STYLEX_CONV_WS=1): measured SLOWER than the per-tile kernels it was meant to replace — 64->64 @256^2, B=64:
    // .54 vs .46 ms forward, .48 vs .41 data gradient; 64->32 .365 vs .351; 32->32 .220 vs .188; whole step 686 vs 699
    // images/s (profiles/r02_c_conv_ws_ab.txt).  With 72 KiB of resident weights only two 20 KiB halo items (40 KiB)
    // can be in flight per CU, against ~76 KiB for two resident blocks of the LDS-DMA kernel: by Little's law the
    // stream tops out near 2 TB/s.  Kept as a correct, tested reference point for that experiment.
    const char* env = getenv("STYLEX_CONV_WS");  // read per launch: the A/B test toggles it in-process
    if (!env || env[0] != '1' || p.mask || p.gate_mask || p.dry) return STYLEX_NOT_APPLICABLE;
    if (!p.act_bf16 || p.s2d_c) return STYLEX_NOT_APPLICABLE;
    if (p.flags & ~(STYLEX_EPI_BIAS | STYLEX_EPI_LRELU | STYLEX_EPI_RELU | STYLEX_EPI_GATE | STYLEX_EPI_OSCALE))
        return STYLEX_NOT_APPLICABLE;
    if ((p.Ck != 32 && p.Ck != 64) || (p.N != 32 && p.N != 64)) return STYLEX_NOT_APPLICABLE;
    if (p.Wo % TW != 0 || p.Ho % TH != 0) return STYLEX_NOT_APPLICABLE;
    const long tiles = (long)p.B * (p.Wo / TW) * (p.Ho / TH);
    if (tiles < 1024) return STYLEX_NOT_APPLICABLE;  // >= 4 tiles per block: the streaming regime
    if ((p.flags & STYLEX_EPI_GATE) && (!p.residual || (reinterpret_cast<uintptr_t>(p.residual) & 15))) return STYLEX_NOT_APPLICABLE;
    if ((p.flags & STYLEX_EPI_OSCALE) && !p.out_scale) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (reinterpret_cast<uintptr_t>(p.y) & 15) ||
        (p.a_scale && (reinterpret_cast<uintptr_t>(p.a_scale) & 15)))
        return STYLEX_NOT_APPLICABLE;
    const int blocks = 256;  // one persistent block per CU
    if (p.N == 64) {
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_ws_kernel<2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WsCfg<2>::SMEM_BYTES);
            if (e != hipSuccess) return (int)e;
            attr = true;
        }
        hipLaunchKernelGGL((conv3x3_ws_kernel<2>), dim3(blocks), dim3(256), WsCfg<2>::SMEM_BYTES, s, p);
    } else {
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_ws_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WsCfg<1>::SMEM_BYTES);
            if (e != hipSuccess) return (int)e;
            attr = true;
        }
        hipLaunchKernelGGL((conv3x3_ws_kernel<1>), dim3(blocks), dim3(256), WsCfg<1>::SMEM_BYTES, s, p);
    }
    return (int)hipGetLastError();
}
