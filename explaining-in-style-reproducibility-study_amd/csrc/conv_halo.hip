// conv_halo.hip — 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) for gfx950 with the
// input tile + halo RESIDENT in LDS: the bf16 speed path of the StylEx step.
//
// Why: in the generic implicit-GEMM kernel (conv_igemm.hip) every one of the 9 taps re-reads its
// 256x32 activation tile from L2 (profiles/r01_a_*: ~2 us per K-tile whatever the MFMA rate).  Here a
// block owns an 8x32 (or 16x16) output-pixel tile of one image; per 32-channel chunk it stages the
// (TH+2)x(TW+2) input halo ONCE (fp32 -> bf16 with v_cvt_pk_bf16_f32 while staging, modulation scale
// folded in) plus the chunk's [BN][9][32] bf16 weights, then issues all 9 taps x 2 k-steps of
// v_mfma_f32_32x32x16_bf16 straight out of LDS: 7x fewer global loads per MFMA, 72 MFMAs per wave
// between barriers.  LDS rows are 80 B (32 bf16 + 8 pad): any 16 rows that differ mod 16 are
// bank-conflict free for ds_read_b128 (slot stride 5).  Next chunk's global loads are issued before
// the MFMAs of the current one (register staging, write after the barrier).
//
// Reference ops replaced: F.conv2d in Conv2DMod.forward (stylex/stylex_train.py:647-667) and the
// 3x3 nn.Conv2d of DiscriminatorBlock (:724-731), plus their input gradients.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);  // v_cvt_pk_bf16_f32 (RNE)
    return *reinterpret_cast<unsigned*>(&r);
}

constexpr int ROWB = 80;  // LDS row pitch in bytes (32 bf16 + 16 B pad)

// ABF: activations are bf16 in HBM -> the halo is staged with raw 16-byte (8-channel) loads and no
// conversion; otherwise fp32 activations are converted to bf16 on the way into LDS.
// S2D: space-to-depth form with per-chunk tap masks (kept out of the plain variant: the mask branches stop the
// compiler from software-pipelining the LDS reads across taps).  EPIX: epilogue may carry noise / residual.
template <int TW, int TN, bool ABF, bool S2D, bool EPIX>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_bf16_kernel(ConvKParams p) {
    constexpr int TH = 256 / TW, HWD = TW + 2, NP = (TH + 2) * HWD;
    constexpr int BN = TN * 32;
    constexpr int SPP = ABF ? 4 : 8;                     // staging slots per halo pixel (32 channels)
    constexpr int SLOT_SHIFT = ABF ? 2 : 3;
    constexpr int HALO_SLOTS = (NP * SPP + 255) / 256;   // global loads per thread per chunk
    constexpr int W_ROWS = BN * 9;
    constexpr int W_SLOTS = (W_ROWS * 4 + 255) / 256;    // 16-byte global loads per thread per chunk
    constexpr int W_OFF = NP * ROWB;
    // tap planes are skewed by 64 B: the staging store puts 4 consecutive taps of one output channel in 16
    // neighbouring lanes, and BN*ROWB is a multiple of the 256-byte bank period
    constexpr int W_TAP = BN * ROWB + 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Ho, W = p.Wo, C = p.Ck;  // stride 1, pad 1: input and output share H, W

    int bid = blockIdx.x;
    {
        int nblk = gridDim.x, q = nblk >> 3, rr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int n_tiles = (p.N + BN - 1) / BN;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int n0 = (bid % n_tiles) * BN;
    int pt = bid / n_tiles;
    const int b = pt / (tiles_x * tiles_y);
    pt -= b * tiles_x * tiles_y;
    const int y0 = (pt / tiles_x) * TH, x0 = (pt % tiles_x) * TW;

    // per-thread halo slots: pixel index in the image batch, or -1 for padding / beyond the halo
    const int q8 = tid & (SPP - 1);
    int hoff[HALO_SLOTS];
#pragma unroll
    for (int it = 0; it < HALO_SLOTS; ++it) {
        int hp = (tid + 256 * it) >> SLOT_SHIFT;
        int hh = hp / HWD, ww = hp - hh * HWD;
        int y = y0 - 1 + hh, x = x0 - 1 + ww;
        bool ok = hp < NP && y >= 0 && y < H && x >= 0 && x < W;
        hoff[it] = ok ? (b * H + y) * W + x : -1;
    }

    // space-to-depth form of a stride-2 conv: which weight taps are structurally non-zero depends on the
    // sub-position of the source-channel chunk (forward) or of this block's output-channel tile (dgrad)
    const unsigned all_taps = 0x1ffu;
    auto wmask_of_chunk = [&](int c0) -> unsigned {
        if (!S2D) return all_taps;
        return stylex_s2d_tap_mask(p.flip_taps ? n0 / p.s2d_c : c0 / p.s2d_c);
    };
    unsigned cur_wmask = all_taps;

    float4 hreg[HALO_SLOTS];  // fp32 x4, or (ABF) raw bf16 x8 bit patterns
    uint4 wreg[W_SLOTS];
    const unsigned short* wsrc = reinterpret_cast<const unsigned short*>(p.w);

    auto issue_loads = [&](int c0) {
        const unsigned wm_next = wmask_of_chunk(c0);
        const int cc = c0 + q8 * (ABF ? 8 : 4);
        const bool cok = cc < C;
#pragma unroll
        for (int it = 0; it < HALO_SLOTS; ++it) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cok && hoff[it] >= 0) {
                if (ABF)  // raw bits: unpacking here would force the load to complete before the MFMAs
                    v = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(p.a) + (long)hoff[it] * C + cc);
                else
                    v = *reinterpret_cast<const float4*>(p.a + (long)hoff[it] * C + cc);
            }
            hreg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < W_SLOTS; ++it) {
            int idx = tid + 256 * it;
            int r = idx >> 2, qq = idx & 3;
            int nl = r / 9, tap = r - nl * 9;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            int ck = c0 + qq * 8;
            int gt = p.flip_taps ? 8 - tap : tap;
            if (r < W_ROWS && n0 + nl < p.N && ck < C && (!S2D || ((wm_next >> gt) & 1))) {
                v = *reinterpret_cast<const uint4*>(wsrc + ((long)(n0 + nl) * 9 + gt) * C + ck);
            }
            wreg[it] = v;
        }
    };

    auto write_lds = [&](int c0) {
        const int cc = c0 + q8 * (ABF ? 8 : 4);
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sc2 = sc;
        if (p.a_scale && cc < C) {
            sc = *reinterpret_cast<const float4*>(p.a_scale + (long)b * C + cc);
            if (ABF) sc2 = *reinterpret_cast<const float4*>(p.a_scale + (long)b * C + cc + 4);
        }
#pragma unroll
        for (int it = 0; it < HALO_SLOTS; ++it) {
            int hp = (tid + 256 * it) >> SLOT_SHIFT;
            if (hp < NP) {
                if (ABF) {
                    uint4 v = make_uint4(__float_as_uint(hreg[it].x), __float_as_uint(hreg[it].y),
                                         __float_as_uint(hreg[it].z), __float_as_uint(hreg[it].w));
                    if (p.a_scale) {
                        float4 f0 = act_unpack4(make_uint2(v.x, v.y)), f1 = act_unpack4(make_uint2(v.z, v.w));
                        v.x = pack_bf16(f0.x * sc.x, f0.y * sc.y); v.y = pack_bf16(f0.z * sc.z, f0.w * sc.w);
                        v.z = pack_bf16(f1.x * sc2.x, f1.y * sc2.y); v.w = pack_bf16(f1.z * sc2.z, f1.w * sc2.w);
                    }
                    *reinterpret_cast<uint4*>(smem + hp * ROWB + q8 * 16) = v;
                } else {
                    uint2 v;
                    v.x = pack_bf16(hreg[it].x * sc.x, hreg[it].y * sc.y);
                    v.y = pack_bf16(hreg[it].z * sc.z, hreg[it].w * sc.w);
                    *reinterpret_cast<uint2*>(smem + hp * ROWB + q8 * 8) = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < W_SLOTS; ++it) {
            int idx = tid + 256 * it;
            int r = idx >> 2, qq = idx & 3;
            if (r < W_ROWS) {
                int nl = r / 9, tap = r - nl * 9;
                *reinterpret_cast<uint4*>(smem + W_OFF + tap * W_TAP + nl * ROWB + qq * 16) = wreg[it];
            }
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // operand base addresses: MFMA row i-tile = 32 consecutive tile pixels, lane li <-> pixel
    const int li = lane & 31, lk = lane >> 5;
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int pix = wave * 64 + i * 32 + li;
        int ph = pix / TW, pw = pix - ph * TW;
        abase[i] = (ph * HWD + pw) * ROWB + lk * 16;
    }
    const int bbase = W_OFF + li * ROWB + lk * 16;

    auto compute = [&]() {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (S2D && !((cur_wmask >> (p.flip_taps ? 8 - tap : tap)) & 1)) continue;  // block-uniform skip
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 av[2], bv[TN];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    av[i] = *reinterpret_cast<const bf16x8*>(smem + abase[i] + (kh * HWD + kw) * ROWB + ks * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bv[j] = *reinterpret_cast<const bf16x8*>(smem + bbase + tap * W_TAP + j * 32 * ROWB + ks * 32);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = ABF ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j], av[i], acc[i][j], 0, 0, 0)   // D^T
                                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    const int nchunks = (C + 31) / 32;
    issue_loads(0);
    write_lds(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) issue_loads((ch + 1) * 32);
        cur_wmask = wmask_of_chunk(ch * 32);
        compute();
        __syncthreads();
        if (more) {
            write_lds((ch + 1) * 32);
            __syncthreads();
        }
    }

    // ---- epilogue.  D[i][j]: col j = lane&31 -> output channel, rows -> 16 tile pixels per lane, i.e. the
    // natural store is one element per lane (2-4 B).  The tile is therefore transposed through LDS (the halo /
    // weight space is dead after the last barrier) and written with 16-byte row stores: a wave covers 1 KiB
    // of contiguous NHWC output when N == BN.
    const int lj = lane & 31, lh = lane >> 5;
    if (ABF) {
        // bf16 activations: the MFMA operands are swapped (D^T = W x X^T), so a lane holds ONE PIXEL (lj) and, per
        // 32-channel sub-tile, the channels 8g + 4lh + (0..3), g = r >> 2: the per-pixel terms (noise) are one value per
        // lane, the per-channel ones float4s, and the store needs no LDS transpose — one v_permlane32_swap per dword
        // pairs the half-waves' quads into 8 consecutive channels (see conv_halo_dma.hip)
        const bool act = (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) != 0;
        const float slope = (p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f;
        const bool w8 = p.N % 8 == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0;
        unsigned short* yout = reinterpret_cast<unsigned short*>(p.y);
        const unsigned short* aux = reinterpret_cast<const unsigned short*>(p.residual);  // residual or gate tensor (bf16)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = wave * 64 + i * 32 + lj;
            const int ph = pix / TW, pw = pix - ph * TW;
            const int y = y0 + ph, x = x0 + pw;
            const bool pix_ok = y < H && x < W;
            const long obase = ((long)(b * H + y) * W + x) * p.N;
            float nz = 0.f;
            if (EPIX && (p.flags & STYLEX_EPI_NOISE)) {
                const int yc = min(y, (int)p.noise_stride - 1), xc = min(x, (int)p.noise_stride - 1);
                nz = (p.flags & STYLEX_EPI_NOISE_NAT) ? p.noise[((long)b * p.noise_stride + yc) * p.noise_stride + xc]
                                                      : p.noise[((long)b * p.noise_stride + xc) * p.noise_stride + yc];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                unsigned P[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + j * 32 + 8 * g + 4 * lh;
                    float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    // N % 4 == 0 on this path (checked by the launcher), n % 4 == 0: the quad is in range as a whole
                    const bool nok = n < p.N;
                    auto ld4 = [&](const float* ptr, float dflt) -> float4 {
                        return nok ? *reinterpret_cast<const float4*>(ptr + n) : make_float4(dflt, dflt, dflt, dflt);
                    };
                    if (p.flags & STYLEX_EPI_OSCALE) {
                        const float4 o4 = ld4(p.out_scale + (long)b * p.N, 1.f);
                        v[0] *= o4.x; v[1] *= o4.y; v[2] *= o4.z; v[3] *= o4.w;
                    }
                    if (p.flags & STYLEX_EPI_BIAS) {
                        const float4 b4 = ld4(p.bias, 0.f);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    }
                    if (EPIX && (p.flags & STYLEX_EPI_NOISE)) {
                        const float4 w4 = ld4(p.noise_w, 0.f), c4 = ld4(p.noise_b, 0.f);
                        v[0] += nz * w4.x + c4.x; v[1] += nz * w4.y + c4.y; v[2] += nz * w4.z + c4.z; v[3] += nz * w4.w + c4.w;
                    }
                    if (EPIX && (p.flags & (STYLEX_EPI_RESIDUAL | STYLEX_EPI_GATE)) && nok && pix_ok) {
                        const uint2 rv = *reinterpret_cast<const uint2*>(aux + obase + n);
                        const float a4[4] = {__uint_as_float(rv.x << 16), __uint_as_float(rv.x & 0xffff0000u),
                                             __uint_as_float(rv.y << 16), __uint_as_float(rv.y & 0xffff0000u)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (p.flags & STYLEX_EPI_RESIDUAL) v[e] = (v[e] + a4[e]) * p.res_scale;
                            else v[e] = a4[e] > 0.f ? v[e] : p.res_scale * v[e];
                        }
                    }
                    if (act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
                    }
                    P[g][0] = pack_bf16(v[0], v[1]);
                    P[g][1] = pack_bf16(v[2], v[3]);
                }
                if (w8) {
#pragma unroll
                    for (int g = 0; g < 4; g += 2)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            auto r2 = __builtin_amdgcn_permlane32_swap(P[g][h], P[g + 1][h], false, false);
                            P[g][h] = r2[0];
                            P[g + 1][h] = r2[1];
                        }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int n = n0 + j * 32 + 16 * q + 8 * lh;
                        if (pix_ok && n < p.N)
                            *reinterpret_cast<uint4*>(yout + obase + n) = make_uint4(P[2 * q][0], P[2 * q][1], P[2 * q + 1][0], P[2 * q + 1][1]);
                    }
                } else {  // odd channel counts (padded RGB outputs): element stores
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + j * 32 + 8 * g + 4 * lh;
                        const unsigned short h4[4] = {(unsigned short)(P[g][0] & 0xffffu), (unsigned short)(P[g][0] >> 16),
                                                      (unsigned short)(P[g][1] & 0xffffu), (unsigned short)(P[g][1] >> 16)};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (pix_ok && n + e < p.N) yout[obase + n + e] = h4[e];
                    }
                }
            }
        }
        return;
    }
    constexpr int OUT_ES = ABF ? 2 : 4;                    // output element size
    constexpr int OROW = BN * OUT_ES + 16;                 // LDS row pitch of the staged output tile (+16 B skew)
    static_assert(256 * OROW <= NP * ROWB + 9 * W_TAP, "output tile must fit in the staging LDS");
    const bool wide = (p.N % (16 / OUT_ES) == 0);          // 16-byte global stores need aligned rows
    if (wide) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + j * 32 + lj;
            const bool nok = n < p.N;
            const float bias = (nok && (p.flags & STYLEX_EPI_BIAS)) ? p.bias[n] : 0.f;
            const float osc = (nok && (p.flags & STYLEX_EPI_OSCALE)) ? p.out_scale[(long)b * p.N + n] : 1.f;
            float nw = 0.f, nb = 0.f;
            if (nok && (p.flags & STYLEX_EPI_NOISE)) {
                nw = p.noise_w[n];
                nb = p.noise_b[n];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // natural-order noise plane: the 4 accumulator rows r = 4q .. 4q+3 are 4 consecutive pixels of one image
                // row -> one aligned 16-byte load each (the transposed plane needs 16 strided 4-byte loads)
                float nz[16];
                if (EPIX && (p.flags & STYLEX_EPI_NOISE) && (p.flags & STYLEX_EPI_NOISE_NAT)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int pix0 = wave * 64 + i * 32 + 8 * q + 4 * lh;
                        const int ph = pix0 / TW, pw = pix0 - ph * TW;
                        const int y = min(y0 + ph, p.noise_stride - 1), x = min(x0 + pw, p.noise_stride - 4);
                        const float4 t = *reinterpret_cast<const float4*>(p.noise + ((long)b * p.noise_stride + y) * p.noise_stride + x);
                        nz[4 * q] = t.x;
                        nz[4 * q + 1] = t.y;
                        nz[4 * q + 2] = t.z;
                        nz[4 * q + 3] = t.w;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pix = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[i][j][r] * osc + bias;
                    if (EPIX && (p.flags & STYLEX_EPI_NOISE) && (p.flags & STYLEX_EPI_NOISE_NAT)) v += nz[r] * nw + nb;
                    if (EPIX && (p.flags & (STYLEX_EPI_NOISE | STYLEX_EPI_RESIDUAL | STYLEX_EPI_GATE))) {
                        const int ph = pix / TW, pw = pix - ph * TW;
                        const int y = min(y0 + ph, H - 1), x = min(x0 + pw, W - 1);
                        if ((p.flags & STYLEX_EPI_NOISE) && !(p.flags & STYLEX_EPI_NOISE_NAT))
                            v += p.noise[((long)b * p.noise_stride + x) * p.noise_stride + y] * nw + nb;
                        if ((p.flags & STYLEX_EPI_RESIDUAL) && nok)
                            v = (v + act_ld1(p.residual, ((long)(b * H + y) * W + x) * p.N + n, p.act_bf16)) * p.res_scale;
                        if ((p.flags & STYLEX_EPI_GATE) && nok)
                            v = act_ld1(p.residual, ((long)(b * H + y) * W + x) * p.N + n, p.act_bf16) > 0.f ? v : p.res_scale * v;
                    }
                    if (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) v = v > 0.f ? v : ((p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f * v);
                    char* d = smem + pix * OROW + (j * 32 + lj) * OUT_ES;
                    if (ABF) *reinterpret_cast<unsigned short*>(d) = (unsigned short)(pack_bf16(v, 0.f) & 0xffffu);
                    else *reinterpret_cast<float*>(d) = v;
                }
            }
        }
        __syncthreads();
        constexpr int CPR = BN * OUT_ES / 16;  // 16-byte chunks per pixel row
#pragma unroll
        for (int k = 0; k < CPR; ++k) {
            const int id = tid + 256 * k;
            const int pix = id / CPR, q = id - pix * CPR;
            const int ph = pix / TW, pw = pix - ph * TW;
            const int y = y0 + ph, x = x0 + pw;
            const int n = n0 + q * (16 / OUT_ES);
            if (y < H && x < W && n < p.N) {
                uint4 v = *reinterpret_cast<const uint4*>(smem + pix * OROW + q * 16);
                char* dst = reinterpret_cast<char*>(p.y) + (((long)(b * H + y) * W + x) * p.N + n) * OUT_ES;
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + lj;
        if (n >= p.N) continue;
        const float bias = (p.flags & STYLEX_EPI_BIAS) ? p.bias[n] : 0.f;
        const float osc = (p.flags & STYLEX_EPI_OSCALE) ? p.out_scale[(long)b * p.N + n] : 1.f;
        float nw = 0.f, nb = 0.f;
        if (p.flags & STYLEX_EPI_NOISE) {
            nw = p.noise_w[n];
            nb = p.noise_b[n];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int pix = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                int ph = pix / TW, pw = pix - ph * TW;
                int y = y0 + ph, x = x0 + pw;
                if (y >= H || x >= W) continue;
                long o = ((long)(b * H + y) * W + x) * p.N + n;
                float v = acc[i][j][r] * osc;
                if (p.flags & STYLEX_EPI_BIAS) v += bias;
                if (p.flags & STYLEX_EPI_NOISE)
                    v += ((p.flags & STYLEX_EPI_NOISE_NAT) ? p.noise[((long)b * p.noise_stride + y) * p.noise_stride + x]
                                                           : p.noise[((long)b * p.noise_stride + x) * p.noise_stride + y]) * nw + nb;
                if (p.flags & STYLEX_EPI_RESIDUAL) v = (v + act_ld1(p.residual, o, p.act_bf16)) * p.res_scale;
                if (p.flags & STYLEX_EPI_GATE) v = act_ld1(p.residual, o, p.act_bf16) > 0.f ? v : p.res_scale * v;
                if (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) v = v > 0.f ? v : ((p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f * v);
                act_st1(p.y, o, v, p.act_bf16);
            }
        }
    }
}

template <int TW, int TN, bool ABF, bool S2D, bool EPIX>
int launch_halo(const ConvKParams& p, hipStream_t s) {
    constexpr int TH = 256 / TW, NP = (TH + 2) * (TW + 2), BN = TN * 32;
    constexpr size_t sm = (size_t)NP * ROWB + 9 * ((size_t)BN * ROWB + 64);
    auto k = conv3x3_halo_bf16_kernel<TW, TN, ABF, S2D, EPIX>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    long tiles = (long)p.B * ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    long blocks = tiles * ((p.N + BN - 1) / BN);
    stylex_note_kernel("conv3x3_halo_bf16_kernel<%d, %d, %s, %s, %s>", TW, TN, ABF ? "true" : "false", S2D ? "true" : "false",
                       EPIX ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), sm, s, p);
    return (int)hipGetLastError();
}

}  // namespace

// Returns STYLEX_NOT_APPLICABLE when the shape is not covered (caller falls back to conv_igemm).
int stylex_launch_halo(const ConvKParams& p, hipStream_t s) {
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1) return STYLEX_NOT_APPLICABLE;
    if (p.Hi != p.Ho || p.Wi != p.Wo) return STYLEX_NOT_APPLICABLE;
    // images below 16 x 16: 12-15 pixel wide ones (the frozen networks' 14 x 14 / 15 x 15 layers, round 6) run here as one partial
    // 16 x 16 tile per image — the generic kernel served them at 100-200 TF/s; <= 8 x 8 belongs to the gather kernel
    static const int min_w = getenv("STYLEX_HALO_MIN_W") ? atoi(getenv("STYLEX_HALO_MIN_W")) : 12;
    if (p.Ck % 8 != 0 || p.Wo < min_w || p.Ho < 8 || (long)p.Ho * p.Wo < (long)min_w * min_w) return STYLEX_NOT_APPLICABLE;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)) return STYLEX_NOT_APPLICABLE;
    if (p.a_scale && (reinterpret_cast<uintptr_t>(p.a_scale) & 15)) return STYLEX_NOT_APPLICABLE;
    {
        int rc = stylex_launch_line64(p, s);  // 64 -> 64 channels at >= 128^2: whole-line DMA, weights resident in LDS
        if (rc != STYLEX_NOT_APPLICABLE) return rc;
        rc = stylex_launch_pipe(p, s);
        if (rc != STYLEX_NOT_APPLICABLE) return rc;
        rc = stylex_launch_halo_dma(p, s);
        if (rc != STYLEX_NOT_APPLICABLE) return rc;
    }
    if (p.mask || p.gate_mask || p.dry) return STYLEX_NOT_APPLICABLE;  // bit masks: LDS-DMA / RGB kernels only
    if (p.act_bf16) {  // the register epilogue reads the per-channel terms as float4 quads
        if (p.N % 4 != 0) return STYLEX_NOT_APPLICABLE;
        const void* quads[4] = {p.bias, p.out_scale, p.noise_w, p.noise_b};
        for (const void* q : quads)
            if (q && (reinterpret_cast<uintptr_t>(q) & 15)) return STYLEX_NOT_APPLICABLE;
        if ((p.flags & (STYLEX_EPI_RESIDUAL | STYLEX_EPI_GATE)) && (reinterpret_cast<uintptr_t>(p.residual) & 7)) return STYLEX_NOT_APPLICABLE;
    }
    const bool wide = p.Wo >= 32;
    const bool epix = (p.flags & (STYLEX_EPI_NOISE | STYLEX_EPI_RESIDUAL | STYLEX_EPI_GATE)) != 0;
    if (p.act_bf16) {
        if (p.s2d_c) {  // always N >= 64 here (s2d_c % 64 == 0)
            if (p.N > 32) return wide ? launch_halo<32, 2, true, true, true>(p, s) : launch_halo<16, 2, true, true, true>(p, s);
            return wide ? launch_halo<32, 1, true, true, true>(p, s) : launch_halo<16, 1, true, true, true>(p, s);
        }
        if (epix) {
            if (p.N > 32) return wide ? launch_halo<32, 2, true, false, true>(p, s) : launch_halo<16, 2, true, false, true>(p, s);
            return wide ? launch_halo<32, 1, true, false, true>(p, s) : launch_halo<16, 1, true, false, true>(p, s);
        }
        if (p.N > 32) return wide ? launch_halo<32, 2, true, false, false>(p, s) : launch_halo<16, 2, true, false, false>(p, s);
        return wide ? launch_halo<32, 1, true, false, false>(p, s) : launch_halo<16, 1, true, false, false>(p, s);
    }
    if (p.s2d_c) {
        if (p.N > 32) return wide ? launch_halo<32, 2, false, true, true>(p, s) : launch_halo<16, 2, false, true, true>(p, s);
        return wide ? launch_halo<32, 1, false, true, true>(p, s) : launch_halo<16, 1, false, true, true>(p, s);
    }
    if (p.N > 32) return wide ? launch_halo<32, 2, false, false, true>(p, s) : launch_halo<16, 2, false, false, true>(p, s);
    return wide ? launch_halo<32, 1, false, false, true>(p, s) : launch_halo<16, 1, false, false, true>(p, s);
}
