// conv_rgb.hip — the FIRST convolution of the discriminator / encoder: 3x3 / stride-1 / pad-1, RGB input padded to one
// 16-byte slot (8 bf16 channels), 64 output channels, bias + LeakyReLU (reference DiscriminatorBlock.net[0],
// stylex_train.py:726-727, on [B,3,256,256]).
//
// The layer is pure HBM streaming (16 B in, 128 B out per pixel; 29 GFLOP at B = 128) and ran at 2.1 TB/s on the generic
// implicit-GEMM kernel.  Here the im2col view is free: K = 9 taps x 8 channels = 72, so the 8 K-values of one MFMA lane
// (k = 8 * (2 s + lane/32) .. + 7) are exactly the 8 channels of ONE tap of ONE halo pixel — a single 16-byte LDS read,
// no gather arithmetic.  Block = 8 x 32 pixel tile (one wave per two rows) x 64 channels: 5.3 KiB halo + 10 KiB weights
// of LDS, so 8 blocks are resident per CU and their loads overlap.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned short to_bf16(float v) {
    f32x2_t t = {v, 0.f};
    bf16x2_t r = __builtin_convertvector(t, bf16x2_t);
    return (unsigned short)(*reinterpret_cast<unsigned*>(&r) & 0xffffu);
}

constexpr int TW = 32, TH = 8, HWD = TW + 2, NP = (TH + 2) * HWD;  // 340 halo pixels of 16 bytes
constexpr int KSTEPS = 5;                                            // 10 tap slots of 8 channels (tap 9 = zeros)
constexpr int W_ROW = KSTEPS * 16 * 2;                               // 160 bytes per output channel: [tap 0..9][8 ch]
constexpr int HALO_BYTES = ((NP * 16 + 255) / 256) * 256;            // 5632
constexpr int W_BYTES = 64 * W_ROW;                                  // 10240
constexpr int SMEM_BYTES = HALO_BYTES + W_BYTES > 16384 ? HALO_BYTES + W_BYTES : 16384;  // 16 KiB: the epilogue's
// per-wave 32 px x 64 n bf16 transpose scratch (4 x 4 KiB) reuses the staging space once the MFMAs are done

__global__ __launch_bounds__(256) void conv3x3_rgb_kernel(ConvKParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Ho, W = p.Wo;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int pt = blockIdx.x;
    const int b = pt / (tiles_x * tiles_y);
    pt -= b * tiles_x * tiles_y;
    const int y0 = (pt / tiles_x) * TH, x0 = (pt % tiles_x) * TW;
    const uint4* xs = reinterpret_cast<const uint4*>(p.a);   // one uint4 = the 8 channels of a pixel
    const uint4* ws = reinterpret_cast<const uint4*>(p.w);   // packed [n][tap][8]: one uint4 per (n, tap)

    // halo: 340 pixels, zero outside the image
    for (int hp = tid; hp < NP; hp += 256) {
        const int hh = hp / HWD, ww = hp - hh * HWD;
        const int y = y0 - 1 + hh, x = x0 - 1 + ww;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (y >= 0 && y < H && x >= 0 && x < W) v = xs[(long)(b * H + y) * W + x];
        *reinterpret_cast<uint4*>(smem + hp * 16) = v;
    }
    // weights: [n][10 tap slots][8 ch], slot 9 zero
    for (int s = tid; s < 64 * 10; s += 256) {
        const int n = s / 10, t = s - n * 10;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (t < 9 && n < p.N) v = ws[n * 9 + t];
        *reinterpret_cast<uint4*>(smem + HALO_BYTES + n * W_ROW + t * 16) = v;
    }
    __syncthreads();

    // wave w: tile rows 2w, 2w+1 (two MFMA row-tiles of 32 pixels) x two column tiles of 32 channels
    const int li = lane & 31, lk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int tap = 2 * s + lk;                      // this lane's tap slot (9 = the zero slot)
        const int kh = tap < 9 ? tap / 3 : 0, kw = tap < 9 ? tap - kh * 3 : 0;
        bf16x8 av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hp = (2 * wave + i + kh) * HWD + li + kw;
            av[i] = tap < 9 ? *reinterpret_cast<const bf16x8*>(smem + hp * 16) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(smem + HALO_BYTES + (j * 32 + li) * W_ROW + tap * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
    }

    // epilogue: bias + LeakyReLU, one 32-pixel row at a time through the wave's scratch (the staging space is dead after
    // the barrier), 16-byte row stores (128 B per pixel)
    __syncthreads();
    char* scr = smem + wave * 4096;
    const bool act = (p.flags & STYLEX_EPI_LRELU) != 0;
    unsigned short* yout = reinterpret_cast<unsigned short*>(p.y);
    float bias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + li;
        bias[j] = ((p.flags & STYLEX_EPI_BIAS) && n < p.N) ? p.bias[n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * lk;
                float v = acc[i][j][r] + bias[j];
                if (act) v = v > 0.f ? v : 0.2f * v;
                *reinterpret_cast<unsigned short*>(scr + px * 128 + (j * 32 + li) * 2) = to_bf16(v);
            }
        const int y = y0 + 2 * wave + i;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = lane + 64 * k;  // 32 px x 8 slots of 8 channels
            const int px = id >> 3, q = id & 7;
            const int x = x0 + px;
            if (y < H && x < W && q * 8 < p.N) {
                const uint4 v = *reinterpret_cast<const uint4*>(scr + px * 128 + q * 16);
                const long o = ((long)(b * H + y) * W + x) * p.N + q * 8;
                *reinterpret_cast<uint4*>(yout + o) = v;
                if (p.mask) p.mask[o >> 3] = (unsigned char)stylex_sign_bits8(v);  // STYLEX_EPI_MASK_OUT
            }
        }
    }
}

}  // namespace

// forward only, bf16 activations, C = 8 (padded RGB), N = 64, epilogue bias / LeakyReLU
int stylex_launch_rgb(const ConvKParams& p, hipStream_t s) {
    const char* env = getenv("STYLEX_CONV_RGB");
    if (env && env[0] == '0') return STYLEX_NOT_APPLICABLE;
    if (!p.act_bf16 || p.a_scale || p.s2d_c || p.flip_taps || p.transposed) return STYLEX_NOT_APPLICABLE;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.Hi != p.Ho || p.Wi != p.Wo) return STYLEX_NOT_APPLICABLE;
    if (p.Ck != 8 || p.N != 64 || p.Wo < 32 || p.Ho < 8) return STYLEX_NOT_APPLICABLE;
    if (p.flags & ~(STYLEX_EPI_BIAS | STYLEX_EPI_LRELU | STYLEX_EPI_MASK_OUT)) return STYLEX_NOT_APPLICABLE;
    if (p.gate_mask) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15) || (reinterpret_cast<uintptr_t>(p.y) & 15))
        return STYLEX_NOT_APPLICABLE;
    if (p.dry) return 0;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_rgb_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const long blocks = (long)p.B * ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    stylex_note_kernel("conv3x3_rgb_kernel");
    hipLaunchKernelGGL(conv3x3_rgb_kernel, dim3((unsigned)blocks), dim3(256), SMEM_BYTES, s, p);
    return (int)hipGetLastError();
}
