// conv_gather.hip — implicit-GEMM convolution for the SMALL-SPATIAL layers (<= 8x8 px, 512 -> 512 channels: blocks 5-7
// and final_conv of the discriminator / encoder, initial_conv and blocks 0-1 of the generator; 3x3/s1/p1 and 1x1/s1),
// bf16 NHWC, forward and data gradient.
//
// Those launches are a few GFLOP each; on the generic kernel (conv_igemm.hip: register-staged, ONE K-tile in flight)
// they take 40-130 us whatever their size because the K loop (72 K-tiles for 3x3x512) is a chain of load latencies.
// Here every operand row is fetched by LDS-DMA straight into the MFMA layout — the per-lane global address of a DMA
// piece is free, so the im2col gather (pixel + tap offset, zero page for padding) costs nothing — through a 4-stage
// LDS ring with THREE stages (96 KiB) in flight per block: the K loop runs at the MFMA rate once the pipe is primed.
//
//   C[m][n] = sum_k A[m][k] * W[n][k],  m = (b, oh, ow), k = (tap, c),  A gathered from x[b, oh +- (kh-p), ow +- (kw-p), c]
//
// Block tile 128 (m) x 128 (n), 4 waves of 64 x 64 (2 x 2 MFMA 32x32x16 tiles).  A K stage = 64 channels of one tap = 4
// groups of 16 channels; LDS image of a group: 128 A rows then 128 W rows of 32 bytes (halves swapped by bit 3 of the
// row, as in conv_halo_dma.hip).  Launches that cannot fill the chip are split along K; every launch writes fp32
// partials [ksplit][M][N] and the existing deterministic split-K epilogue kernel (conv_igemm.hip) applies the fused
// epilogue (scales, bias, noise, residual, gate, activation) and the bf16 rounding.
//
// Round 5: conv_gather_line_kernel (below) — same tile, split-K plan and partial layout as the round-2 kernel it replaced, operand
// rows staged as whole 128-byte lines on 8 waves and, above all, a DMA issue path without branches, 64-bit pointer arithmetic or
// kernel-argument reloads inside the loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr int GBM = 128, GBN = 128;
constexpr int RING = 4;

struct GatherParams {
    const unsigned short* x;   // [B][H][W][C] bf16
    const unsigned short* w;   // [N][T][C] bf16 (forward pack, or data-gradient pack)
    float* partial;            // [ksplit][M][N]
    unsigned short* y;         // != nullptr: ksplit == 1 and no epilogue — store bf16 [M][N] directly, no partials
    int B, H, W, C, N, KH, pad, sign;  // H, W: output grid.  sign +1: forward gather ih = oh*stride + kh - p;
                                       // -1: data gradient ih = (oh + p - kh) / stride
    int Hs, Ws, stride;                // source grid; stride 1, or 2 (the 3x3/p1 down conv of a DiscriminatorBlock)
    int m_tiles;                       // M tiles of the launch (stride-2 data gradient: 4 parity classes x tiles of a class)
    int M, stages, ksplit, stages_per_split;
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short f2bf_rne(float f) {  // v_cvt_pk_bf16_f32, as the split-K epilogue rounds
    f32x2_t v = {f, 0.f};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return (unsigned short)(*reinterpret_cast<unsigned*>(&r) & 0xffffu);
}

// ---- WHOLE-LINE staging on 8 waves (round 5).  The round-2 kernel (deleted in round 6; git history, profiles/design_history_r1_r4.md)
// staged 16-channel groups: every DMA instruction fetched 32 bytes from each of 32 rows, a request shape the vector-memory path serves at 15-17 B/clk/CU whatever
// is in flight (tools/l2_feed_probe.hip), and with the ring no longer drained at every barrier that rate IS its stage time
// (32 KiB per stage = 1.0 us for 0.25 us of MFMA work).  Here a K stage is the same 64 channels of one tap, but an operand row
// is ONE 128-byte line in global memory and in LDS: a DMA instruction fetches 8 complete lines (33-37 B/clk/CU with 8 waves
// issuing), 16-byte slot q of row R lives at physical slot q ^ ((R >> 1) & 7) (conv_line64.hip: an operand read of 16
// consecutive rows and one logical slot covers all 64 banks).  Waves 0-3 stage the A rows (gather), waves 4-7 the W rows;
// wave (wm, wn) = (wave >> 1, wave & 1) owns a 32 (m) x 64 (n) part of the 128 x 128 tile.
constexpr int LROW = 128;                        // bytes of an operand row: 64 channels
constexpr int LSTAGE = (GBM + GBN) * LROW;       // 32 KiB
constexpr int LPIECES = 4;                       // DMA instructions per wave and stage (8 rows x 128 B each)

__device__ __forceinline__ void lds16(bf16x8& dst, int addr) {  // (asm: the compiler must not wait for it right away)
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}

__global__ __launch_bounds__(512, 1) void conv_gather_line_kernel(GatherParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = (p.N + GBN - 1) / GBN;
    const int n0 = (int)(blockIdx.x % n_tiles) * GBN;
    int mt = (int)(blockIdx.x / n_tiles);
    const bool phased = p.sign < 0 && p.stride == 2;  // stride-2 data gradient: rows grouped by output parity class (above)
    int py = 0, px = 0;
    if (phased) {
        const int tpp = p.m_tiles >> 2;
        const int ph = mt / tpp;
        mt -= ph * tpp;
        py = ph >> 1;
        px = ph & 1;
    }
    const int m0 = mt * GBM;
    const int rows = phased ? p.B * p.Hs * p.Ws : p.M;
    const int cpt = p.C >> 6;  // stages per tap
    const int ks = blockIdx.y;
    const int s_begin = ks * p.stages_per_split;
    const int s_end = min(s_begin + p.stages_per_split, phased ? (1 + py) * (1 + px) * cpt : p.stages);
    const int nst = max(s_end - s_begin, 0);
    const int T = p.KH * p.KH, K = T * p.C;

    // ---- staging role of this lane: DMA instruction `it` of a stage = rows (wave & 3) * 32 + it * 8 + (lane >> 3) of the A
    // panel (waves 0-3) or of the W panel (waves 4-7); physical slot lane & 7 holds logical slot (lane & 7) ^ swz(row).
    // Everything a stage's DMA needs per lane is a byte offset computed from registers: buffer loads with an out-of-range
    // offset for padding / missing rows (zeros in LDS), the tap and channel-chunk terms as the scalar offset.  (The first
    // version computed 64-bit pointers per piece through branches and re-read kernel arguments from memory inside the loop:
    // eight scalar-load round trips per stage in front of every wave's MFMAs.)
    const bool stage_w = wave >= 4;
    const int rbase = (wave & 3) * 32 + (lane >> 3);
    const int Hs = __builtin_amdgcn_readfirstlane(p.Hs), Ws = __builtin_amdgcn_readfirstlane(p.Ws);
    const int C = __builtin_amdgcn_readfirstlane(p.C), KH = __builtin_amdgcn_readfirstlane(p.KH);
    const int pad = __builtin_amdgcn_readfirstlane(p.pad), sgn = __builtin_amdgcn_readfirstlane(p.sign);
    const int strd = __builtin_amdgcn_readfirstlane(p.stride);
    const int gw = phased ? Ws : p.W;
    const int hw = phased ? Hs * Ws : p.H * p.W;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.w), 0, 0x7ffffff0, 0x00020000);
    constexpr unsigned OOBV = 0x80000000u;
    int r_oh[LPIECES], r_ow[LPIECES];
    unsigned r_off[LPIECES];  // A: byte offset of (image, channel slot) or OOBV when the row is past the tile's rows; W: of (n, slot)
#pragma unroll
    for (int it = 0; it < LPIECES; ++it) {
        const int row = rbase + it * 8;
        const unsigned chan = (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 3);
        r_oh[it] = r_ow[it] = 0;
        if (stage_w) {
            const int n = n0 + row;
            r_off[it] = n < p.N ? ((unsigned)n * (unsigned)K + chan) * 2u : OOBV;
        } else {
            const int m = m0 + row;
            if (m < rows) {
                const int b = m / hw, q = m - b * hw;
                r_oh[it] = q / gw;
                r_ow[it] = q - r_oh[it] * gw;
                r_off[it] = ((unsigned)b * (unsigned)(Hs * Ws) * (unsigned)C + chan) * 2u;
            } else {
                r_off[it] = OOBV;
            }
        }
    }
    int i_tap = s_begin / cpt, i_chunk = s_begin - i_tap * cpt;  // the stage the DMA stream fetches next
    auto issue_stage = [&](int rs) {
        int tap = i_tap, dh, dw, sh = 1;  // source pixel = (oh * sh + dh, ow * sh + dw)
        if (phased) {  // tap = index into the class's live taps; (2q + py + 1 - kh) / 2 = q + (py && kh == 0)
            const int nkw = 1 + px;
            const int a = tap / nkw, bq = tap - a * nkw;
            const int kh = py ? 2 * a : 1, kw = px ? 2 * bq : 1;
            dh = (py && kh == 0) ? 1 : 0;
            dw = (px && kw == 0) ? 1 : 0;
            tap = kh * 3 + kw;
        } else {
            const int kh = KH == 3 ? (tap >= 6 ? 2 : tap >= 3 ? 1 : 0) : 0, kw = tap - kh * KH;
            if (sgn > 0) {
                sh = strd;
                dh = kh - pad;
                dw = kw - pad;
            } else {
                dh = pad - kh;
                dw = pad - kw;
            }
        }
        const int dst = rs * LSTAGE + (stage_w ? GBM * LROW : 0) + (wave & 3) * 32 * LROW;
        if (stage_w) {
            const unsigned soff = (unsigned)(tap * C + (i_chunk << 6)) * 2u;
#pragma unroll
            for (int it = 0; it < LPIECES; ++it)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_ptr)(smem + dst + it * 8 * LROW), 16, r_off[it], soff, 0, 0);
        } else {
            const unsigned soff = (unsigned)(i_chunk << 6) * 2u;
#pragma unroll
            for (int it = 0; it < LPIECES; ++it) {
                const int ih = r_oh[it] * sh + dh, iw = r_ow[it] * sh + dw;
                const bool ok = r_off[it] != OOBV && (unsigned)ih < (unsigned)Hs && (unsigned)iw < (unsigned)Ws;
                const unsigned v = ok ? r_off[it] + (unsigned)((ih * Ws + iw) * C) * 2u : OOBV;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_ptr)(smem + dst + it * 8 * LROW), 16, v, soff, 0, 0);
            }
        }
        if (++i_chunk == cpt) {
            i_chunk = 0;
            ++i_tap;
        }
    };

    // ---- MFMA operand rows: A rows wm * 32 + li, W rows wn * 64 + j * 32 + li; (row >> 1) & 7 == (li >> 1) & 7 for all of them
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int s0 = lk ^ ((li >> 1) & 7);  // logical slot 2 g + lk of k-step g sits at physical slot (2 g) ^ s0
    const int lds0 = (int)(unsigned)(uintptr_t)(lds_void_ptr)smem;  // LDS byte address of the dynamic allocation
    const int a_off = lds0 + (wm * 32 + li) * LROW;
    const int b_off = lds0 + GBM * LROW + (wn * 64 + li) * LROW;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // prologue: three stages in flight
    for (int s = 0; s < 3 && s < nst; ++s) issue_stage(s);
    for (int s = 0; s < nst; ++s) {
        // stage s must have landed; loads complete in order, so allowing the LPIECES DMA instructions of each younger stage in
        // flight proves it
        if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPIECES) : "memory");
        else if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // (raw: __syncthreads() would drain the ring)
        asm volatile("" ::: "memory");
        if (s + 3 < nst) issue_stage((s + 3) % RING);
        // operand reads run one 16-channel group ahead of the MFMAs (the per-group read -> wait -> MFMA chain of the kernel
        // above exposes the LDS latency four times per stage: with one or two waves per SIMD that chain WAS the stage time)
        const int base = (s % RING) * LSTAGE;
        bf16x8 av[2], bv[2][2];
        lds16(av[0], base + a_off + (s0 << 4));
        lds16(bv[0][0], base + b_off + (s0 << 4));
        lds16(bv[0][1], base + b_off + 32 * LROW + (s0 << 4));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
                const int so = ((2 * (g + 1)) ^ s0) << 4;
                lds16(av[(g + 1) & 1], base + a_off + so);
                lds16(bv[(g + 1) & 1][0], base + b_off + so);
                lds16(bv[(g + 1) & 1][1], base + b_off + 32 * LROW + so);
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(av[g & 1]), "+v"(bv[g & 1][0]), "+v"(bv[g & 1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[g & 1]), "+v"(bv[g & 1][0]), "+v"(bv[g & 1][1]));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g & 1], bv[g & 1][j], acc[j], 0, 0, 0);
        }
    }

    // raw fp32 partials: D[row = m][col = n], col = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float* out = p.partial + (long)ks * p.M * p.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int mm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const bool ok = mm < rows && n < p.N;
            if (phased && ok) {  // class row -> natural (b, oh, ow) order of the partial buffer
                const int bb = mm / hw, qq = mm - bb * hw;
                const int qh = qq / p.Ws, qw = qq - qh * p.Ws;
                mm = (bb * p.H + 2 * qh + py) * p.W + 2 * qw + px;
            }
            if (ok && p.y) p.y[(long)mm * p.N + n] = f2bf_rne(acc[j][r]);
            else if (ok) out[(long)mm * p.N + n] = acc[j][r];
        }
    }
}

}  // namespace

// Plan: K stages of 64 channels; split along K until ~256 blocks exist, at least 4 stages per slice.
static bool gather_applicable(const ConvKParams& p) {
    const char* env = getenv("STYLEX_CONV_GATHER");
    if (env && env[0] == '0') return false;
    if (env && env[0] == 's' && p.stride != 1) return false;  // "s1": stride-1 layers only (A/B of the stride-2 mode)
    if (!p.act_bf16 || p.a_scale || p.s2d_c) return false;
    if (p.KH != p.KW || !((p.KH == 3 && p.pad == 1) || (p.KH == 1 && p.pad == 0))) return false;
    if (p.stride == 1) {
        if (p.Hi != p.Ho || p.Wi != p.Wo || p.phase_major) return false;
    } else {  // 3x3/s2/p1: forward Hi = 2 Ho; data gradient (ConvKParams names the dy grid Hi, the dx grid Ho) Ho = 2 Hi
        if (p.stride != 2 || p.KH != 3) return false;
        if (p.transposed ? (p.Ho != 2 * p.Hi || p.Wo != 2 * p.Wi) : (p.Hi != 2 * p.Ho || p.Wi != 2 * p.Wo)) return false;
    }
    if ((p.transposed && p.stride == 2 ? p.Hi * p.Wi : p.Ho * p.Wo) > 64) return false;
    if (p.Ck % 64 != 0 || p.N % 8 != 0 || p.N < 64) return false;
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)) return false;
    return true;
}

static int gather_m_tiles(const ConvKParams& p) {
    if (p.transposed && p.stride == 2) return 4 * ((p.B * p.Hi * p.Wi + GBM - 1) / GBM);
    return (p.M + GBM - 1) / GBM;
}

// K stages of the longest tile (stride-2 data gradient: the 4-tap parity class)
static int gather_stages(const ConvKParams& p) {
    return (p.transposed && p.stride == 2 ? 4 : p.KH * p.KW) * (p.Ck / 64);
}

static void gather_plan(const ConvKParams& p, int* ksplit, int* per) {
    const int stages = gather_stages(p);
    const long tiles = (long)gather_m_tiles(p) * ((p.N + GBN - 1) / GBN);
    int ks = (int)((256 + tiles - 1) / tiles);
    if (ks > stages / 4) ks = stages / 4;
    if (ks < 1) ks = 1;
    int pp = (stages + ks - 1) / ks;
    *per = pp;
    *ksplit = (stages + pp - 1) / pp;
}

int64_t stylex_gather_workspace_bytes(const ConvKParams& p) {
    if (!gather_applicable(p)) return 0;
    int ks, per;
    gather_plan(p, &ks, &per);
    return (int64_t)ks * p.M * p.N * (int64_t)sizeof(float);
}

// Fills p.ksplit / p.partial for the split-K epilogue the caller launches afterwards.  STYLEX_NOT_APPLICABLE when the
// shape is not covered or the workspace is too small.
int stylex_launch_gather(ConvKParams& p, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    if (!gather_applicable(p) || !workspace) return STYLEX_NOT_APPLICABLE;
    int ks, per;
    gather_plan(p, &ks, &per);
    if (workspace_bytes < (int64_t)ks * p.M * p.N * (int64_t)sizeof(float)) return STYLEX_NOT_APPLICABLE;
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return STYLEX_NOT_APPLICABLE;
    // 0 = not asked yet, 1 = granted, -1 = refused (a device with less LDS): the generic kernel behind this one serves the launch
    static int attr_state = 0;
    if (attr_state == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gather_line_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, RING * LSTAGE);
        attr_state = e == hipSuccess ? 1 : -1;
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (attr_state < 0) return STYLEX_NOT_APPLICABLE;
    GatherParams g;
    g.x = reinterpret_cast<const unsigned short*>(p.a);
    g.w = reinterpret_cast<const unsigned short*>(p.w);
    g.partial = (float*)workspace;
    // one K slice and nothing to fuse (a plain data gradient): the kernel rounds and stores the result itself
    const bool direct = ks == 1 && p.flags == 0 && !p.out_scale && !p.bias;
    g.y = direct ? reinterpret_cast<unsigned short*>(p.y) : nullptr;
    g.B = p.B;
    g.H = p.Ho;
    g.W = p.Wo;
    g.C = p.Ck;
    g.N = p.N;
    g.KH = p.KH;
    g.pad = p.pad;
    g.sign = p.transposed ? -1 : 1;
    g.Hs = p.Hi;
    g.Ws = p.Wi;
    g.stride = p.stride;
    g.m_tiles = gather_m_tiles(p);
    g.M = p.M;
    g.stages = gather_stages(p);
    g.ksplit = ks;
    g.stages_per_split = per;
    const long tiles = (long)g.m_tiles * ((p.N + GBN - 1) / GBN);
    stylex_note_kernel("conv_gather_line_kernel");
    hipLaunchKernelGGL(conv_gather_line_kernel, dim3((unsigned)tiles, (unsigned)ks), dim3(512), RING * LSTAGE, s, g);
    p.ksplit = direct ? 0 : ks;  // 0: output complete, the caller skips the split-K epilogue
    p.kt_per_split = per;
    p.partial = (float*)workspace;
    return (int)hipGetLastError();
}
