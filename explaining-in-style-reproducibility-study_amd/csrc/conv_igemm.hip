// conv_igemm.hip — implicit-GEMM convolution kernels for gfx950 (MI355X, CDNA4).
//
// One LDS-tiled MFMA kernel family serves the three convolution roles of the
// StylEx train step (reference call sites: F.conv2d in Conv2DMod.forward,
// stylex/stylex_train.py:647-667; nn.Conv2d in DiscriminatorBlock :724-744,
// Generator.initial_conv :771, DiscriminatorE.final_conv :881, and the
// convolution_backward / double-backward nodes autograd derives from them):
//
//   fprop   y[m][n]  = sum_k  gatherF(x)[m][k] * Wf[n][k]        m=(b,ho,wo) k=(tap,c)
//   dgrad   dx[m][c] = sum_k  gatherT(dy)[m][k] * Wb[c][k]       m=(b,hi,wi) k=(tap,n)
//   wgrad   dW[n][tap][c] = sum_m dy[m][n] * gatherF(x)[m][tap][c]
//
// Activations are NHWC fp32, so every gathered row is a contiguous channel
// vector: loads are 16-byte, 128-byte-coalesced per 8 lanes.  Tiles are staged
// through LDS k-major ([k][row], row stride padded by one dword) so that the
// fp32 MFMA operand read (one dword per lane, lanes 0-31 = 32 consecutive rows)
// is bank-conflict free, and the transposing store is conflict free too.
// Register-staged double buffering: the global loads of K-tile t+1 are issued
// before the MFMAs of tile t and written to the other LDS stage afterwards
// (one barrier per K-tile).
//
// Two arithmetic modes:
//   F32  : v_mfma_f32_32x32x2_f32 — bitwise an fmaf chain, parity mode.
//   BF16 : v_mfma_f32_32x32x16_bf16 — operands rounded (RNE) to bf16 when staged
//          into LDS, fp32 accumulate.
//
// Modulation (Conv2DMod) never materialises per-sample weights: the per-sample
// input-channel scale (style+1) is applied to the gathered operand while it is
// staged, and the demodulation coefficient in the epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stylex_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 32;  // K-tile depth (fp32 elements)

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);  // v_cvt_pk_bf16_f32, round-to-nearest-even
    return *reinterpret_cast<unsigned*>(&r);
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(pack_bf16(f, 0.f) & 0xffffu); }

// ------------------------------------------------------------------------------------------
// Row descriptor: which gathered source rows (pixels) a thread stages.
// ------------------------------------------------------------------------------------------
struct RowPix {
    int b, oh, ow;  // output-space pixel of GEMM row m (valid==false => zero row)
    bool valid;
};

__device__ __forceinline__ RowPix decode_row(const ConvKParams& p, int m) {
    RowPix r;
    r.valid = m < p.M;
    int mm = r.valid ? m : 0;
    if (p.phase_major) {
        // transposed stride-2 gather: rows ordered (phase_h, phase_w, b, oh/2, ow/2) so that a tile
        // shares the parity class and the taps that cannot contribute are skipped tile-uniformly.
        int hh = p.Ho >> 1, wh = p.Wo >> 1;
        int per = p.B * hh * wh;
        int ph = mm / per;
        int rem = mm - ph * per;
        r.b = rem / (hh * wh);
        int q = rem - r.b * hh * wh;
        r.oh = 2 * (q / wh) + (ph >> 1);
        r.ow = 2 * (q % wh) + (ph & 1);
    } else {
        int hw = p.Ho * p.Wo;
        r.b = mm / hw;
        int q = mm - r.b * hw;
        r.oh = q / p.Wo;
        r.ow = q - r.oh * p.Wo;
    }
    return r;
}

// source pixel of (row, tap) -> element offset of its channel vector, or -1 when it is padding
__device__ __forceinline__ long src_offset(const ConvKParams& p, const RowPix& r, int kh, int kw) {
    if (!r.valid) return -1;
    int ih, iw;
    if (!p.transposed) {
        ih = r.oh * p.stride + kh - p.pad;
        iw = r.ow * p.stride + kw - p.pad;
    } else {
        int th = r.oh + p.pad - kh, tw = r.ow + p.pad - kw;
        if (th < 0 || tw < 0) return -1;
        if (p.stride == 2) {
            if ((th | tw) & 1) return -1;
            ih = th >> 1;
            iw = tw >> 1;
        } else {
            ih = th;
            iw = tw;
        }
    }
    if (ih < 0 || iw < 0 || ih >= p.Hi || iw >= p.Wi) return -1;
    return ((long)(r.b * p.Hi + ih) * p.Wi + iw) * p.Ck;
}

// ------------------------------------------------------------------------------------------
// fprop / dgrad kernel.  Block = 256 threads = WM x WN waves, wave tile (TM*32) x (TN*32).
// ------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, bool VEC4, bool BF16, int BKT, bool ABF = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvKParams p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // ABF: bf16 activations AND weights staged as raw 16-byte slots of 8 elements (no conversion, no unpack)
    constexpr int KQ = ABF ? BKT / 8 : BKT / 4;  // slots per staged row
    constexpr int RPP = 256 / KQ;    // rows staged per pass of the 256 threads
    constexpr int RA = BM / RPP, RB = (BN + RPP - 1) / RPP;  // rows staged per thread for A / B
    constexpr bool B_PARTIAL = (BN % RPP) != 0;                // RPP = 64 with a 32-column tile
    // fp32 LDS image: [k][row] with row stride +1 dword.  bf16 image: [row][k] bf16, 80-byte rows.
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr int LDH = BKT + 8;  // bf16 elements per LDS row (+8 pad: 16-B aligned rows, odd 16-B slot stride)
    constexpr int A_ELEMS = BF16 ? (BM * LDH / 2) : (BKT * LDA);
    constexpr int B_ELEMS = BF16 ? (BN * LDH / 2) : (BKT * LDB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto As = [&](int st) -> float* { return smem + st * (A_ELEMS + B_ELEMS); };
    auto Bs = [&](int st) -> float* { return smem + st * (A_ELEMS + B_ELEMS) + A_ELEMS; };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: consecutive logical tiles (sharing activation rows) stay on one XCD's L2.
    int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int n_tiles = (p.N + BN - 1) / BN;
    const int m0 = (bid / n_tiles) * BM;
    const int n0 = (bid % n_tiles) * BN;

    const int kq = tid % KQ;  // which float4 of the BKT-deep K chunk
    const int r0 = tid / KQ;  // 0..RPP-1

    RowPix rows[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) rows[j] = decode_row(p, m0 + r0 + RPP * j);

    const int T = p.KH * p.KW;
    // K-tiles: VEC4 -> (tap, 32-channel chunk); scalar -> flattened k = tap*Ck + c
    const int chunks = (p.Ck + BKT - 1) / BKT;
    // Ck == 4 (RGB padded to 4): tap-major flattening k = tap*4 + c, each float4 slot is one tap
    const bool c4 = VEC4 && (ABF ? p.Ck == 8 : p.Ck == 4);
    const int nk = VEC4 ? (c4 ? (T + KQ - 1) / KQ : T * chunks) : (T * p.Ck + BKT - 1) / BKT;

    // tile-uniform tap skipping for the phase-major transposed stride-2 gather
    int skip_ph = -1, skip_pw = -1;
    if (p.phase_major) {
        int per = p.B * (p.Ho >> 1) * (p.Wo >> 1);
        int last = min(m0 + BM, p.M) - 1;
        if (m0 / per == last / per) {
            int ph = m0 / per;
            skip_ph = ((ph >> 1) + p.pad) & 1;  // taps with (kh&1) != skip_ph never hit
            skip_pw = ((ph & 1) + p.pad) & 1;
        }
    }

    float4 ra[RA], rb[RB];
    uint4 rah[RA], rbh[RB];  // ABF staging registers (raw bf16 x8)
    int cur_kt = 0;          // K-tile held in the staging registers (ABF scale lookup)

    auto load_tile = [&](int kt) {
        if (ABF) {
            cur_kt = kt;
            const unsigned short* abase = reinterpret_cast<const unsigned short*>(p.a);
            const unsigned short* wbase = reinterpret_cast<const unsigned short*>(p.w);
            int tap = c4 ? kt * KQ + kq : kt / chunks;
            int c0 = c4 ? 0 : (kt - tap * chunks) * BKT + kq * 8;
            int kh = tap / p.KW, kw = tap - kh * p.KW;
            bool cok = c4 ? tap < T : c0 < p.Ck;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                long off = cok ? src_offset(p, rows[j], kh, kw) : -1;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (off >= 0) v = *reinterpret_cast<const uint4*>(abase + off + c0);
                rah[j] = v;
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                int n = n0 + r0 + RPP * j;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (cok && n < p.N && (!B_PARTIAL || r0 + RPP * j < BN))
                    v = *reinterpret_cast<const uint4*>(wbase + ((long)n * T + tap) * p.Ck + c0);
                rbh[j] = v;
            }
            return;
        }
        if (VEC4) {
            int tap = c4 ? kt * KQ + kq : kt / chunks;
            int c0 = c4 ? 0 : (kt - tap * chunks) * BKT + kq * 4;
            int kh = tap / p.KW, kw = tap - kh * p.KW;
            bool cok = c4 ? tap < T : c0 < p.Ck;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                long off = cok ? src_offset(p, rows[j], kh, kw) : -1;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (off >= 0) {
                    v = act_ld4(p.a, off + c0, p.act_bf16);
                    if (p.a_scale) {
                        float4 s = *reinterpret_cast<const float4*>(p.a_scale + (long)rows[j].b * p.Ck + c0);
                        v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
                    }
                }
                ra[j] = v;
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                int n = n0 + r0 + RPP * j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cok && n < p.N) {
                    long wo = ((long)n * T + tap) * p.Ck + c0;
                    if (BF16) {  // weights are pre-packed bf16: 4 elements = 8 bytes, kept packed in rb[j].xy
                        uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.w) + wo);
                        v.x = __uint_as_float(h.x);
                        v.y = __uint_as_float(h.y);
                    } else {
                        v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.w) + wo);
                    }
                }
                rb[j] = v;
            }
        } else {
            int kbase = kt * BKT + kq * 4;
            int KT = T * p.Ck;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = kbase + e;
                    v[e] = 0.f;
                    if (k < KT) {
                        int tap = k / p.Ck, c = k - tap * p.Ck;
                        int kh = tap / p.KW, kw = tap - kh * p.KW;
                        long off = src_offset(p, rows[j], kh, kw);
                        if (off >= 0) {
                            v[e] = act_ld1(p.a, off + c, p.act_bf16);
                            if (p.a_scale) v[e] *= p.a_scale[(long)rows[j].b * p.Ck + c];
                        }
                    }
                }
                ra[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                int n = n0 + r0 + RPP * j;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int k = kbase + e;
                    v[e] = 0.f;
                    if (k < KT && n < p.N) {
                        if (BF16)
                            v[e] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p.w)[(long)n * KT + k] << 16);
                        else
                            v[e] = reinterpret_cast<const float*>(p.w)[(long)n * KT + k];
                    }
                }
                if (BF16)  // same packed form as the vector path
                    rb[j] = make_float4(__uint_as_float(pack_bf16(v[0], v[1])), __uint_as_float(pack_bf16(v[2], v[3])), 0.f, 0.f);
                else
                    rb[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };

    auto store_tile = [&](int st) {
        if (ABF) {
            unsigned short* a = reinterpret_cast<unsigned short*>(As(st));
            unsigned short* b = reinterpret_cast<unsigned short*>(Bs(st));
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                uint4 v = rah[j];
                if (p.a_scale && rows[j].valid) {  // modulation scale: unpack, scale, repack
                    int c0 = c4 ? 0 : (cur_kt % chunks) * BKT + kq * 8;
                    if (c0 < p.Ck) {
                        const float* sp = p.a_scale + (long)rows[j].b * p.Ck + c0;
                        float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
                        float4 f0 = act_unpack4(make_uint2(v.x, v.y)), f1 = act_unpack4(make_uint2(v.z, v.w));
                        v.x = pack_bf16(f0.x * s0.x, f0.y * s0.y); v.y = pack_bf16(f0.z * s0.z, f0.w * s0.w);
                        v.z = pack_bf16(f1.x * s1.x, f1.y * s1.y); v.w = pack_bf16(f1.z * s1.z, f1.w * s1.w);
                    }
                }
                *reinterpret_cast<uint4*>(a + (r0 + RPP * j) * LDH + kq * 8) = v;
            }
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (!B_PARTIAL || r0 + RPP * j < BN) *reinterpret_cast<uint4*>(b + (r0 + RPP * j) * LDH + kq * 8) = rbh[j];
            return;
        }
        if (BF16) {
            unsigned short* a = reinterpret_cast<unsigned short*>(As(st));
            unsigned short* b = reinterpret_cast<unsigned short*>(Bs(st));
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                uint2 v;
                v.x = pack_bf16(ra[j].x, ra[j].y);
                v.y = pack_bf16(ra[j].z, ra[j].w);
                *reinterpret_cast<uint2*>(a + (r0 + RPP * j) * LDH + kq * 4) = v;
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                uint2 v;
                v.x = __float_as_uint(rb[j].x);
                v.y = __float_as_uint(rb[j].y);
                *reinterpret_cast<uint2*>(b + (r0 + RPP * j) * LDH + kq * 4) = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                float* d = As(st) + (kq * 4) * LDA + r0 + RPP * j;
                d[0] = ra[j].x; d[LDA] = ra[j].y; d[2 * LDA] = ra[j].z; d[3 * LDA] = ra[j].w;
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                float* d = Bs(st) + (kq * 4) * LDB + r0 + RPP * j;
                d[0] = rb[j].x; d[LDB] = rb[j].y; d[2 * LDB] = rb[j].z; d[3 * LDB] = rb[j].w;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto tile_skipped = [&](int kt) -> bool {
        if (skip_ph < 0 || !VEC4 || c4) return false;
        int tap = kt / chunks;
        int kh = tap / p.KW, kw = tap - kh * p.KW;
        return ((kh & 1) != skip_ph) || ((kw & 1) != skip_pw);
    };

    auto compute = [&](int st) {
        const int li = lane & 31, lk = lane >> 5;
        if (BF16) {
            const unsigned short* a = reinterpret_cast<const unsigned short*>(As(st));
            const unsigned short* b = reinterpret_cast<const unsigned short*>(Bs(st));
#pragma unroll
            for (int ks = 0; ks < BKT / 16; ++ks) {
                bf16x8 av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    av[i] = *reinterpret_cast<const bf16x8*>(a + (wm * TM * 32 + i * 32 + li) * LDH + ks * 16 + lk * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bv[j] = *reinterpret_cast<const bf16x8*>(b + (wn * TN * 32 + j * 32 + li) * LDH + ks * 16 + lk * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const float* a = As(st) + wm * TM * 32 + li;
            const float* b = Bs(st) + wn * TN * 32 + li;
#pragma unroll
            for (int ks = 0; ks < BKT / 2; ++ks) {
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = a[(ks * 2 + lk) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = b[(ks * 2 + lk) * LDB + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop over the non-skipped K-tiles, double buffered
    // split-K: blockIdx.y selects a contiguous slice of K-tiles; partial sums go to p.partial
    const int kslice = blockIdx.y;
    int kt = kslice * p.kt_per_split;
    const int kt_end = p.ksplit > 1 ? min(nk, kt + p.kt_per_split) : nk;
    while (kt < kt_end && tile_skipped(kt)) ++kt;
    int st = 0;
    if (kt < kt_end) {
        load_tile(kt);
        store_tile(0);
    }
    __syncthreads();
    while (kt < kt_end) {
        int nxt = kt + 1;
        while (nxt < kt_end && tile_skipped(nxt)) ++nxt;
        if (nxt < kt_end) load_tile(nxt);
        compute(st);
        if (nxt < kt_end) store_tile(st ^ 1);
        __syncthreads();
        st ^= 1;
        kt = nxt;
    }

    // ---- epilogue: D[i][j], col j = lane&31 (output channel), row i = (r&3)+8*(r>>2)+4*(lane>>5) (pixel)
    // Rows are ordered (b, oh, ow) except in the phase-major gather, so the NHWC output offset of row m is m * N: the
    // (b, oh, ow) decomposition (two integer divisions) is only needed for the per-image scale and the noise plane.
    // It used to run for each of the 64 accumulator elements of a lane and was the ~40 us floor of every small launch.
    const int lj = lane & 31, lh = lane >> 5;
    const bool split = p.ksplit > 1;
    const bool need_pix = p.phase_major || (!split && (p.flags & (STYLEX_EPI_OSCALE | STYLEX_EPI_NOISE)));
    int nn[TN];
    float bias[TN], nw[TN], nb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        nn[j] = n0 + wn * TN * 32 + j * 32 + lj;
        bias[j] = nw[j] = nb[j] = 0.f;
        if (nn[j] < p.N && !split) {
            if (p.flags & STYLEX_EPI_BIAS) bias[j] = p.bias[nn[j]];
            if (p.flags & STYLEX_EPI_NOISE) {
                nw[j] = p.noise_w[nn[j]];
                nb[j] = p.noise_b[nn[j]];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= p.M) continue;
            RowPix rp;
            rp.b = rp.oh = rp.ow = 0;
            long orow = (long)m * p.N;
            if (need_pix) {
                rp = decode_row(p, m);
                orow = ((long)(rp.b * p.Ho + rp.oh) * p.Wo + rp.ow) * p.N;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nn[j];
                if (n >= p.N) continue;
                const long o = orow + n;
                float v = acc[i][j][r];
                if (split) {  // raw slice sum; the epilogue runs in splitk_epilogue_kernel
                    p.partial[(long)kslice * p.M * p.N + o] = v;
                    continue;
                }
                if (p.flags & STYLEX_EPI_OSCALE) v *= p.out_scale[(long)rp.b * p.N + n];
                if (p.flags & STYLEX_EPI_BIAS) v += bias[j];
                if (p.flags & STYLEX_EPI_NOISE)
                    v += ((p.flags & STYLEX_EPI_NOISE_NAT) ? p.noise[((long)rp.b * p.noise_stride + rp.oh) * p.noise_stride + rp.ow]
                                                           : p.noise[((long)rp.b * p.noise_stride + rp.ow) * p.noise_stride + rp.oh]) * nw[j] + nb[j];
                if (p.flags & STYLEX_EPI_RESIDUAL) v = (v + act_ld1(p.residual, o, p.act_bf16)) * p.res_scale;
                if (p.flags & STYLEX_EPI_GATE) v = act_ld1(p.residual, o, p.act_bf16) > 0.f ? v : p.res_scale * v;
                if (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) v = v > 0.f ? v : ((p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f * v);
                act_st1(p.y, o, v, p.act_bf16);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad kernel: partial[split][n][tap][c] = sum over this split's pixels of dy[m][n]*x_gather[m][tap][c]
// Block tile 128 (n) x 128 (c) per (tap, split): 4 waves as 2x2, wave tile 64x64.
// LDS image is k-major straight from memory ([pixel][channel]): no transpose needed for fp32.
// ------------------------------------------------------------------------------------------
template <int TN_, int TC_, bool VEC4, bool BF16>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvKParams p) {
    constexpr int WNn = 2, WNc = 2;
    constexpr int BNn = WNn * TN_ * 32, BC = WNc * TC_ * 32;
    constexpr int LDA = BNn + 4, LDB = BC + 4;
    constexpr int LDHA = BK + 8, LDHB = BK + 8;  // bf16 image [row][k], rows = channel, k = pixel
    constexpr int A_ELEMS = BF16 ? (BNn * LDHA / 2) : (BK * LDA);
    constexpr int B_ELEMS = BF16 ? (BC * LDHB / 2) : (BK * LDB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto As = [&](int st) -> float* { return smem + st * (A_ELEMS + B_ELEMS); };
    auto Bs = [&](int st) -> float* { return smem + st * (A_ELEMS + B_ELEMS) + A_ELEMS; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int T = p.KH * p.KW;
    const int n_tiles = (p.N + BNn - 1) / BNn, c_tiles = (p.Ck + BC - 1) / BC;

    int bid = blockIdx.x;
    const int tile = bid % (n_tiles * c_tiles * T);
    const int split = bid / (n_tiles * c_tiles * T);
    const int tap = tile % T;
    const int n0 = ((tile / T) / c_tiles) * BNn;
    const int c0 = ((tile / T) % c_tiles) * BC;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;

    const long m_begin = (long)split * p.split_len;
    const long m_end = min((long)p.M, m_begin + p.split_len);
    const int nk = (int)((m_end - m_begin + BK - 1) / BK);

    // staging: each K-tile = 32 pixels; A row (pixel) = BNn floats of dy, B row = BC floats of x.
    constexpr int A_V = BNn / 4, B_V = BC / 4;            // float4 per pixel row
    constexpr int A_PER = (BK * A_V) / 256, B_PER = (BK * B_V) / 256;
    float4 ra[A_PER], rb[B_PER];

    auto load_tile = [&](int kt) {
        long mb = m_begin + (long)kt * BK;
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            int idx = tid + 256 * j;
            int pr = idx / A_V, q = idx % A_V;
            long m = mb + pr;
            int n = n0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                const long src = m * p.N + n;
                int bb = (int)(m / ((long)p.Ho * p.Wo));
                if (VEC4) {
                    if (n < p.N) {
                        v = act_ld4(p.a2, src, p.act_bf16);
                        if (p.a2_scale) {
                            float4 s = *reinterpret_cast<const float4*>(p.a2_scale + (long)bb * p.N + n);
                            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
                        }
                    }
                } else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t[e] = (n + e < p.N) ? act_ld1(p.a2, src + e, p.act_bf16) : 0.f;
                        if (p.a2_scale && n + e < p.N) t[e] *= p.a2_scale[(long)bb * p.N + n + e];
                    }
                    v = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            int idx = tid + 256 * j;
            int pr = idx / B_V, q = idx % B_V;
            long m = mb + pr;
            int c = c0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                RowPix rp = decode_row(p, (int)m);
                long off = src_offset(p, rp, kh, kw);
                if (off >= 0) {
                    if (VEC4) {
                        if (c < p.Ck) {
                            v = act_ld4(p.a, off + c, p.act_bf16);
                            if (p.a_scale) {
                                float4 s = *reinterpret_cast<const float4*>(p.a_scale + (long)rp.b * p.Ck + c);
                                v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
                            }
                        }
                    } else {
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            t[e] = (c + e < p.Ck) ? act_ld1(p.a, off + c + e, p.act_bf16) : 0.f;
                            if (p.a_scale && c + e < p.Ck) t[e] *= p.a_scale[(long)rp.b * p.Ck + c + e];
                        }
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
            rb[j] = v;
        }
    };

    auto store_tile = [&](int st) {
        if (BF16) {
            // transpose while storing: image [channel][pixel] bf16 so the MFMA operand read is 16 B
            unsigned short* a = reinterpret_cast<unsigned short*>(As(st));
            unsigned short* b = reinterpret_cast<unsigned short*>(Bs(st));
#pragma unroll
            for (int j = 0; j < A_PER; ++j) {
                int idx = tid + 256 * j;
                int pr = idx / A_V, q = idx % A_V;
                a[(q * 4 + 0) * LDHA + pr] = f2bf(ra[j].x);
                a[(q * 4 + 1) * LDHA + pr] = f2bf(ra[j].y);
                a[(q * 4 + 2) * LDHA + pr] = f2bf(ra[j].z);
                a[(q * 4 + 3) * LDHA + pr] = f2bf(ra[j].w);
            }
#pragma unroll
            for (int j = 0; j < B_PER; ++j) {
                int idx = tid + 256 * j;
                int pr = idx / B_V, q = idx % B_V;
                b[(q * 4 + 0) * LDHB + pr] = f2bf(rb[j].x);
                b[(q * 4 + 1) * LDHB + pr] = f2bf(rb[j].y);
                b[(q * 4 + 2) * LDHB + pr] = f2bf(rb[j].z);
                b[(q * 4 + 3) * LDHB + pr] = f2bf(rb[j].w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_PER; ++j) {
                int idx = tid + 256 * j;
                int pr = idx / A_V, q = idx % A_V;
                *reinterpret_cast<float4*>(As(st) + pr * LDA + q * 4) = ra[j];
            }
#pragma unroll
            for (int j = 0; j < B_PER; ++j) {
                int idx = tid + 256 * j;
                int pr = idx / B_V, q = idx % B_V;
                *reinterpret_cast<float4*>(Bs(st) + pr * LDB + q * 4) = rb[j];
            }
        }
    };

    f32x16 acc[TN_][TC_];
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TC_; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int st) {
        const int li = lane & 31, lk = lane >> 5;
        if (BF16) {
            const unsigned short* a = reinterpret_cast<const unsigned short*>(As(st));
            const unsigned short* b = reinterpret_cast<const unsigned short*>(Bs(st));
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 av[TN_], bv[TC_];
#pragma unroll
                for (int i = 0; i < TN_; ++i)
                    av[i] = *reinterpret_cast<const bf16x8*>(a + (wn * TN_ * 32 + i * 32 + li) * LDHA + ks * 16 + lk * 8);
#pragma unroll
                for (int j = 0; j < TC_; ++j)
                    bv[j] = *reinterpret_cast<const bf16x8*>(b + (wc * TC_ * 32 + j * 32 + li) * LDHB + ks * 16 + lk * 8);
#pragma unroll
                for (int i = 0; i < TN_; ++i)
#pragma unroll
                    for (int j = 0; j < TC_; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const float* a = As(st) + wn * TN_ * 32 + li;
            const float* b = Bs(st) + wc * TC_ * 32 + li;
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                float av[TN_], bv[TC_];
#pragma unroll
                for (int i = 0; i < TN_; ++i) av[i] = a[(ks * 2 + lk) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TC_; ++j) bv[j] = b[(ks * 2 + lk) * LDB + j * 32];
#pragma unroll
                for (int i = 0; i < TN_; ++i)
#pragma unroll
                    for (int j = 0; j < TC_; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0, st = 0; kt < nk; ++kt, st ^= 1) {
        if (kt + 1 < nk) load_tile(kt + 1);
        compute(st);
        if (kt + 1 < nk) store_tile(st ^ 1);
        __syncthreads();
    }

    // partial[split][n][tap][c]
    const int lj = lane & 31, lh = lane >> 5;
    float* out = p.y + (long)split * p.N * T * p.Ck;
#pragma unroll
    for (int j = 0; j < TC_; ++j) {
        int c = c0 + wc * TC_ * 32 + j * 32 + lj;
        if (c >= p.Ck) continue;
#pragma unroll
        for (int i = 0; i < TN_; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = n0 + wn * TN_ * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < p.N) out[((long)n * T + tap) * p.Ck + c] = acc[i][j][r];
            }
    }
}

// Bias gradient riding the weight-gradient reduce launch (round 5; its own launch was 22 us of serial dependent loads,
// 27 times per step): the blocks behind the `main_blocks` reduce blocks sum the per-split pixel sums of dy the
// weight-gradient kernel left in bias_part[split][n].  One wave per channel, lane l takes splits l, l + 64, ..., then a
// fixed shuffle tree: deterministic.
// dw = (acc ? dw : 0) + scale * sum, as two rounded fp32 operations: bit-identical to the sum stored, multiplied in place
// and added by the autograd engine (the launches this replaces)
__device__ __forceinline__ float wg_out(float sum, float old, float scale, int acc) {
#pragma clang fp contract(off)  // (hipcc contracts a * b + c into one fma by default — one rounding instead of two)
    float v = scale * sum;
    asm volatile("" : "+v"(v));
    return acc ? old + v : v;
}

__device__ __forceinline__ void bias_reduce_block(const float* __restrict__ part, float* __restrict__ db, int N, int splits,
                                                  int blk, float scale, int acc) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blk * 4 + w;
    if (n >= N) return;
    float v = 0.f;
    for (int s = l; s < splits; s += 64) v += part[(long)s * N + n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (l == 0) db[n] = wg_out(v, acc ? db[n] : 0.f, scale, acc);
}

// dw_oihw[n][c][t] = sum_split partial[split][n][t][c]   (fixed order => deterministic)
// One thread owns one (n, c): for each tap it reads the slices with lanes running along c (coalesced
// 256-byte rows), four slices in flight per tap.  The T results of a thread are consecutive in OIHW, so a block's
// 256 pairs form one contiguous span of 256*T floats: it is transposed through LDS and written with coalesced
// rows (the direct form wrote 4 bytes at a 36-byte lane stride).  Fixed summation order -> deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* dw,
                                                           int N, int C, int T, int splits, int main_blocks,
                                                           const float* __restrict__ bias_part, float* db, float scale, int accum) {
    __shared__ float tile[256 * 9];
    if ((int)blockIdx.x >= main_blocks) {
        bias_reduce_block(bias_part, db, N, splits, blockIdx.x - main_blocks, scale, accum);
        return;
    }
    const long total = (long)N * C * T;
    const long pairs = (long)N * C;
    const long i0 = (long)blockIdx.x * 256;
    const long i = i0 + threadIdx.x;
    if (T == 9) {
        float acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = 0.f;
        if (i < pairs) {
            const int c = (int)(i % C);
            const long n = i / C;
            const float* src = partial + (n * 9) * C + c;
            int k = 0;
            for (; k + 4 <= splits; k += 4) {
                float v[4][9];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 9; ++t) v[u][t] = src[(long)(k + u) * total + (long)t * C];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[t] += v[u][t];
            }
            for (; k < splits; ++k) {
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t] += src[(long)k * total + (long)t * C];
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) tile[threadIdx.x * 9 + t] = acc[t];
        __syncthreads();
        const long span = min((long)256, pairs - i0) * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int o = j * 256 + threadIdx.x;
            if (o < span) dw[i0 * 9 + o] = wg_out(tile[o], accum ? dw[i0 * 9 + o] : 0.f, scale, accum);
        }
        return;
    }
    if (i >= pairs) return;
    const int c = (int)(i % C);
    const long n = i / C;
    const float* src = partial + (n * T) * C + c;
    float* dst = dw + i * T;
    for (int t = 0; t < T; ++t) {
        float a = 0.f;
        for (int k = 0; k < splits; ++k) a += src[(long)k * total + (long)t * C];
        dst[t] = wg_out(a, accum ? dst[t] : 0.f, scale, accum);
    }
}

// Few outputs, many slices (first-layer wgrads: 64 x 8 x 9 outputs from several hundred pixel splits): the
// kernel above would run 2 blocks with 768-long serial sums.  Here a block owns PP (n,c) pairs and G slice
// groups: thread (g, p) sums slices g, g+G, ... of its pair, the G partial sums are combined through LDS in
// fixed order (deterministic), T outputs per pair.
template <int PP, int G>
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ partial, float* dw,
                                                                 int N, int C, int T, int splits, int main_blocks,
                                                                 const float* __restrict__ bias_part, float* db, float scale, int accum) {
    static_assert(PP * G == 256, "one thread per (pair, slice group)");
    __shared__ float red[G][PP][9];
    if ((int)blockIdx.x >= main_blocks) {
        bias_reduce_block(bias_part, db, N, splits, blockIdx.x - main_blocks, scale, accum);
        return;
    }
    const long total = (long)N * C * T;
    const long pairs = (long)N * C;
    const int p = threadIdx.x % PP, g = threadIdx.x / PP;
    const long i = (long)blockIdx.x * PP + p;
    const bool ok = i < pairs;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    if (ok) {
        const float* src = partial + ((i / C) * T) * C + (int)(i % C);
        // four slices in flight per tap (the loop is pure memory latency: PMC showed 95 % of its cycles waiting),
        // added in slice order so the summation order does not depend on the unrolling
        int k = g;
        for (; k + 3 * G < splits; k += 4 * G) {
            float v[4][9];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (t < T) v[u][t] = src[(long)(k + u * G) * total + (long)t * C];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (t < T) acc[t] += v[u][t];
        }
        for (; k < splits; k += G) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if (t < T) acc[t] += src[(long)k * total + (long)t * C];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[g][p][t] = acc[t];
    __syncthreads();
    // PP*T outputs of the block are contiguous in OIHW: thread o sums its G slice-group partials in fixed order
    for (int o = threadIdx.x; o < PP * T; o += 256) {
        const int pp = o / T, t = o - pp * T;
        if ((long)blockIdx.x * PP + pp < pairs) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < G; ++q) sum += red[q][pp][t];
            float* d = dw + (long)blockIdx.x * PP * T + o;
            *d = wg_out(sum, accum ? *d : 0.f, scale, accum);
        }
    }
}

// bias_part / db != null: the same launch also reduces the per-split bias sums (bias_reduce_block)
struct WgOut {  // dw = (accumulate ? dw : 0) + scale * sum
    float scale;
    int accumulate;
};
static WgOut wg_out_of(const ConvKParams& p) { return WgOut{p.wg_scale != 0.f ? p.wg_scale : 1.f, p.wg_accumulate}; }

static void launch_wgrad_reduce(const float* partial, float* dw, int N, int C, int T, int splits, hipStream_t s, WgOut wo,
                                const float* bias_part = nullptr, float* db = nullptr) {
    const long pairs = (long)N * C;
    const unsigned extra = (bias_part && db) ? (unsigned)((N + 3) / 4) : 0u;
    if (T <= 9 && splits >= 32 && (pairs <= 2048 || (pairs <= 8192 && splits >= 256))) {
        const unsigned mb = (unsigned)((pairs + 7) / 8);
        hipLaunchKernelGGL((wgrad_reduce_small_kernel<8, 32>), dim3(mb + extra), dim3(256), 0, s, partial, dw, N, C, T, splits,
                           (int)mb, bias_part, db, wo.scale, wo.accumulate);
    } else if (T <= 9 && splits >= 16 && pairs <= 16384) {
        const unsigned mb = (unsigned)((pairs + 31) / 32);
        hipLaunchKernelGGL((wgrad_reduce_small_kernel<32, 8>), dim3(mb + extra), dim3(256), 0, s, partial, dw, N, C, T, splits,
                           (int)mb, bias_part, db, wo.scale, wo.accumulate);
    } else {
        const unsigned mb = (unsigned)((pairs + 255) / 256);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(mb + extra), dim3(256), 0, s, partial, dw, N, C, T, splits, (int)mb,
                           bias_part, db, wo.scale, wo.accumulate);
    }
}

template <typename OUT>
__device__ __forceinline__ OUT cvt_out(float v);
template <>
__device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <>
__device__ __forceinline__ unsigned short cvt_out<unsigned short>(float v) { return f2bf(v); }

template <typename OUT>
__global__ void pack_weight_kernel(const float* __restrict__ w, OUT* __restrict__ wf, OUT* __restrict__ wb, int N,
                                   int C, int T) {
    long total = (long)N * C * T;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int t = (int)(i % T);
        int c = (int)((i / T) % C);
        int n = (int)(i / ((long)T * C));
        OUT v = cvt_out<OUT>(w[i]);
        if (wf) wf[((long)n * T + t) * C + c] = v;
        if (wb) wb[((long)c * T + t) * N + n] = v;
    }
}

// y[o] = epilogue( sum_s partial[s][o] )   (fixed order => deterministic), o = ((b*Ho+oh)*Wo+ow)*N + n
__device__ __forceinline__ float splitk_finish(const ConvKParams& p, float v, long o, int n, int b, int oh, int ow) {
    if (p.flags & STYLEX_EPI_OSCALE) v *= p.out_scale[(long)b * p.N + n];
    if (p.flags & STYLEX_EPI_BIAS) v += p.bias[n];
    if (p.flags & STYLEX_EPI_NOISE)
        v += ((p.flags & STYLEX_EPI_NOISE_NAT) ? p.noise[((long)b * p.noise_stride + oh) * p.noise_stride + ow]
                                               : p.noise[((long)b * p.noise_stride + ow) * p.noise_stride + oh]) * p.noise_w[n] + p.noise_b[n];
    if (p.flags & STYLEX_EPI_RESIDUAL) v = (v + act_ld1(p.residual, o, p.act_bf16)) * p.res_scale;
    if (p.flags & STYLEX_EPI_GATE) v = act_ld1(p.residual, o, p.act_bf16) > 0.f ? v : p.res_scale * v;
    if (p.flags & (STYLEX_EPI_LRELU | STYLEX_EPI_RELU)) v = v > 0.f ? v : ((p.flags & STYLEX_EPI_RELU) ? 0.f : 0.2f * v);
    return v;
}

// VEC: N % 4 == 0 and M*N < 2^31 — four consecutive channels per thread (16-byte partial loads, one 32-bit division
// per quad, and the (b, oh, ow) decomposition only when a per-image scale or the noise plane asks for it); the
// element-at-a-time 64-bit form paid three 64-bit divisions per output.
template <bool VEC>
__global__ void splitk_epilogue_kernel(ConvKParams p) {
    const long total = (long)p.M * p.N;
    if (VEC) {
        const int tot = (int)total, N = p.N, hw = p.Ho * p.Wo;
        const bool need_pix = p.flags & (STYLEX_EPI_OSCALE | STYLEX_EPI_NOISE);
        for (int o = (blockIdx.x * blockDim.x + threadIdx.x) * 4; o < tot; o += gridDim.x * blockDim.x * 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = 0; s < p.ksplit; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(p.partial + (long)s * total + o);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const int pix = o / N, n = o - pix * N;
            int b = 0, oh = 0, ow = 0;
            if (need_pix) {
                b = pix / hw;
                const int q = pix - b * hw;
                oh = q / p.Wo;
                ow = q - oh * p.Wo;
            }
            float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = splitk_finish(p, r[e], (long)o + e, n + e, b, oh, ow);
            if (p.act_bf16) {
                uint2 h;
                h.x = pack_bf16(r[0], r[1]);
                h.y = pack_bf16(r[2], r[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + o) = h;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + o) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
        return;
    }
    for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += p.partial[(long)s * total + o];
        int n = (int)(o % p.N);
        long pix = o / p.N;
        int hw = p.Ho * p.Wo;
        int b = (int)(pix / hw);
        int q = (int)(pix - (long)b * hw);
        int oh = q / p.Wo, ow = q - oh * p.Wo;
        act_st1(p.y, o, splitk_finish(p, v, o, n, b, oh, ow), p.act_bf16);
    }
}

// Stride-2 3x3/pad-1 conv rewritten over the space-to-depth input (4 sub-positions x C channels):
//   W2[n][(s,c)][kh2][kw2] = W[n][c][kmap(kh2,sy)][kmap(kw2,sx)]  for kh2,kw2 in {0,1} and (kh2==1||sy==1)&&(kw2==1||sx==1)
//   (frame offset -1 <-> kh2 = 0 reads the odd sub-row: original kh = 0; offset 0 <-> kh2 = 1: kh = 1 + sy), else 0.
// Packed bf16 directly into the two operand layouts of a 3x3/s1 conv with Ck = 4C.
__global__ void pack_weight_s2d_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                       unsigned short* __restrict__ wb, int N, int C) {
    const int C4 = 4 * C;
    const long total = (long)N * C4 * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int sc = (int)(i % C4);           // fastest: source channel in s2d space (coalesced wf writes)
        int t2 = (int)((i / C4) % 9);
        int n = (int)(i / ((long)C4 * 9));
        int s = sc / C, c = sc - s * C, sy = s >> 1, sx = s & 1;
        int kh2 = t2 / 3, kw2 = t2 - kh2 * 3;
        float v = 0.f;
        if (kh2 < 2 && kw2 < 2 && (kh2 == 1 || sy == 1) && (kw2 == 1 || sx == 1)) {
            int kh = kh2 == 0 ? 0 : 1 + sy, kw = kw2 == 0 ? 0 : 1 + sx;
            v = w[((long)n * C + c) * 9 + kh * 3 + kw];
        }
        unsigned short h = f2bf(v);
        if (wf) wf[((long)n * 9 + t2) * C4 + sc] = h;
        if (wb) wb[((long)sc * 9 + t2) * N + n] = h;
    }
}

// dW[n][c][kh][kw] = dW2[n][(s,c)][kh2][kw2] for the unique (s, kh2, kw2) that maps onto (kh, kw)
__global__ void fold_weight_s2d_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int N, int C) {
    const long total = (long)N * C * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int t = (int)(i % 9);
        int c = (int)((i / 9) % C);
        long n = i / (9L * C);
        int kh = t / 3, kw = t - kh * 3;
        int kh2 = kh == 0 ? 0 : 1, sy = kh == 0 ? 1 : kh - 1;
        int kw2 = kw == 0 ? 0 : 1, sx = kw == 0 ? 1 : kw - 1;
        dw[i] = dw2[((n * 4 * C) + (sy * 2 + sx) * C + c) * 9 + kh2 * 3 + kw2];
    }
}

template <int WM, int WN, int TM, int TN, bool VEC4, bool BF16, int BKT>
constexpr size_t igemm_smem() {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    return BF16 ? (size_t)2 * (BM + BN) * (BKT + 8) * 2 : (size_t)2 * BKT * (BM + 1 + BN + 1) * 4;
}
template <int TN_, int TC_, bool BF16>
constexpr size_t wgrad_smem() {
    constexpr int BNn = 2 * TN_ * 32, BC = 2 * TC_ * 32;
    return BF16 ? (size_t)2 * (BNn + BC) * (BK + 8) * 2 : (size_t)2 * BK * (BNn + 4 + BC + 4) * 4;
}

template <int WM, int WN, int TM, int TN, bool VEC4, bool BF16, int BKT, bool ABF = false>
int launch_igemm(const ConvKParams& p, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    auto k = conv_igemm_kernel<WM, WN, TM, TN, VEC4, BF16, BKT, ABF>;
    constexpr size_t sm = igemm_smem<WM, WN, TM, TN, VEC4, BF16, BKT>();
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    long blocks = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    stylex_note_kernel("conv_igemm_kernel<%d, %d, %d, %d, %s, %s, %d, %s>", WM, WN, TM, TN, VEC4 ? "true" : "false",
                       BF16 ? "true" : "false", BKT, ABF ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks, (unsigned)(p.ksplit > 1 ? p.ksplit : 1)), dim3(256), sm, s, p);
    return (int)hipGetLastError();
}

// Launches with 4096 <= M rows but at most 256 tiles of 128x128 (the 8x8 px 512->512 layers at B = 64..128) run 64x64
// tiles: 4x the blocks at 37 KB of LDS each -> 4 resident per CU, so the K loop's load latency overlaps across blocks
// (measured at B = 128: forward .165 -> .132 ms, data gradient .181 -> .143).  Below that (4x4 / 2x2 px) the 128x128
// tile with split-K stays: the small tile measured 2x slower there (.054 -> .101 ms).  STYLEX_IGEMM_SMALL=0 disables.
static bool igemm_small_tile(const ConvKParams& p) {
    static const int mode = getenv("STYLEX_IGEMM_SMALL") ? atoi(getenv("STYLEX_IGEMM_SMALL")) : 1;
    if (!mode || p.N <= 64 || p.N % 64 != 0) return false;
    long blocks128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    return (mode == 2 || p.M >= 4096) && blocks128 <= 256;
}

template <bool VEC4, bool BF16, int BKT, bool ABF = false>
int dispatch_igemm(const ConvKParams& p, hipStream_t s) {
    // tile choice by output-channel count
    if (ABF && BKT == 64 && igemm_small_tile(p)) return launch_igemm<2, 2, 1, 1, VEC4, BF16, BKT, ABF>(p, s);
    if (p.N > 64) return launch_igemm<2, 2, 2, 2, VEC4, BF16, BKT, ABF>(p, s);
    if (p.N > 32) return launch_igemm<4, 1, 2, 2, VEC4, BF16, BKT, ABF>(p, s);
    return launch_igemm<4, 1, 2, 1, VEC4, BF16, BKT, ABF>(p, s);
}

template <int TN_, int TC_, bool VEC4, bool BF16>
int launch_wgrad(const ConvKParams& p, int blocks, hipStream_t s) {
    auto k = conv_wgrad_kernel<TN_, TC_, VEC4, BF16>;
    constexpr size_t sm = wgrad_smem<TN_, TC_, BF16>();
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    stylex_note_kernel("conv_wgrad_kernel<%d, %d, %s, %s>", TN_, TC_, VEC4 ? "true" : "false", BF16 ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), sm, s, p);
    return (int)hipGetLastError();
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host-side entry points used by stylex_capi.hip
// ------------------------------------------------------------------------------------------
// tile geometry chosen by dispatch_igemm (must stay in sync with it)
static void igemm_tile(const ConvKParams& p, int* bm, int* bn) {
    if (p.act_bf16 && p.Ck % 8 == 0 && igemm_small_tile(p)) { *bm = 64; *bn = 64; }
    else if (p.N > 64) { *bm = 128; *bn = 128; }
    else if (p.N > 32) { *bm = 256; *bn = 64; }
    else { *bm = 256; *bn = 32; }
}

// K-tile depth: the bf16 path is load-latency bound per K-tile, so deeper tiles (more bytes in flight and
// more MFMAs per barrier) are used whenever the channel count allows
static int igemm_bk(const ConvKParams& p, bool vec, int precision) {
    // bf16 activations: 16-byte slots of 8 elements => 64-deep tiles with the same thread mapping.
    if (precision == STYLEX_BF16 && p.act_bf16 && p.Ck % 8 == 0 && vec)
        return p.N > 64 ? 64 : 33;  // 33 = "32-deep ABF": the 256-row tiles would drop to 1 block/CU at 64
    // fp32 activations: BKT = 64/128 measured SLOWER (the big strided layers are L2/HBM-bound on the fp32
    // gather and lose occupancy, the small ones are launch-bound).  Keep 32.
    return 32;
}

// split-K plan for launches that cannot fill the chip: few output tiles and a long K loop
static void igemm_splitk_plan(const ConvKParams& p, bool vec, int precision, int* ksplit, int* kt_per) {
    int bm, bn;
    igemm_tile(p, &bm, &bn);
    long blocks = (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    int T = p.KH * p.KW;
    int bk = igemm_bk(p, vec, precision);
    const bool abf = bk == 64 || bk == 33;
    if (bk == 33) bk = 32;
    int nk = vec ? ((p.Ck == 4 && !abf) ? (T + 7) / 8 : (abf && p.Ck == 8) ? (T + bk / 8 - 1) / (bk / 8) : T * ((p.Ck + bk - 1) / bk)) : (T * p.Ck + BK - 1) / BK;
    *ksplit = 1;
    *kt_per = nk;
    // Small-spatial layers launch a handful of tiles whose K loop is pure latency (measured: 4-64 blocks, 35-45 us
    // per launch whatever the size): slice K down to 2 K-tiles per block.
    // The fp32 parity mode keeps the coarser plan (>= 4 K-tiles per slice, K loops of >= 16 tiles only): the
    // multi-step loss goldens were pinned with that summation order, and the untrained GAN amplifies a 1e-7
    // reordering to a few 1e-3 within five steps.
    const bool exact = precision == STYLEX_F32;
    if (blocks >= 192 || nk < (exact ? 16 : 4)) return;
    long want = (512 + blocks - 1) / blocks;
    long maxs = nk / (exact ? 4 : 2);  // K-tiles per slice: >= 4 (fp32) / >= 2 (bf16)
    if (want > maxs) want = maxs;
    if (want > 36) want = 36;
    if (want < 2) return;
    int per = (int)((nk + want - 1) / want);
    *kt_per = per;
    *ksplit = (nk + per - 1) / per;
}

static bool igemm_vec_ok(const ConvKParams& p) {
    return (p.Ck % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.a) & 15) == 0) &&
           ((reinterpret_cast<uintptr_t>(p.w) & 15) == 0) &&
           (!p.a_scale || (reinterpret_cast<uintptr_t>(p.a_scale) & 15) == 0);
}

int64_t stylex_igemm_workspace_bytes(const ConvKParams& p, int precision) {
    int ks, per;
    igemm_splitk_plan(p, p.Ck % 4 == 0, precision == STYLEX_F32 ? STYLEX_F32 : STYLEX_BF16, &ks, &per);
    int64_t own = ks > 1 ? (int64_t)ks * p.M * p.N * (int64_t)sizeof(float) : 0;
    int64_t gather = precision == STYLEX_F32 ? 0 : stylex_gather_workspace_bytes(p);
    return own > gather ? own : gather;
}

static int launch_splitk_epilogue(const ConvKParams& p, hipStream_t s) {
    long total = (long)p.M * p.N;
    const bool vec4 = p.N % 4 == 0 && total < (1L << 31) && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.partial) & 15) == 0;
    int blocks = (int)(((vec4 ? total / 4 : total) + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (vec4)
        hipLaunchKernelGGL(splitk_epilogue_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(splitk_epilogue_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

int stylex_launch_igemm(ConvKParams p, int precision, void* workspace, int64_t workspace_bytes, hipStream_t s) {
    bool vec = igemm_vec_ok(p);
    if (precision == STYLEX_BF16 && !p.transposed) {
        int rc = stylex_launch_rgb(p, s);
        if (rc != STYLEX_NOT_APPLICABLE) return rc;
        rc = stylex_launch_halo(p, s);
        if (rc != STYLEX_NOT_APPLICABLE) return rc;
    }
    if (p.mask || p.gate_mask) return STYLEX_EINVAL;  // the caller did not ask stylex_conv_mask_supported()
    p.ksplit = 1;
    p.kt_per_split = 0;
    p.partial = nullptr;
    if (precision == STYLEX_BF16) {  // <= 8x8 px layers: LDS-DMA implicit GEMM + the split-K epilogue kernel
        int rc = stylex_launch_gather(p, workspace, workspace_bytes, s);
        if (rc == 0) return p.ksplit ? launch_splitk_epilogue(p, s) : 0;
        if (rc != STYLEX_NOT_APPLICABLE) return rc;
    }
    if (workspace) {
        int ks, per;
        igemm_splitk_plan(p, vec, precision, &ks, &per);
        if (ks > 1 && workspace_bytes >= (int64_t)ks * p.M * p.N * (int64_t)sizeof(float)) {
            p.ksplit = ks;
            p.kt_per_split = per;
            p.partial = (float*)workspace;
        }
    }
    int rc;
    if (precision == STYLEX_BF16) {
        const int bk = igemm_bk(p, vec, precision);
        if (bk == 64) rc = dispatch_igemm<true, true, 64, true>(p, s);
        else if (bk == 33) rc = dispatch_igemm<true, true, 32, true>(p, s);
        else rc = vec ? dispatch_igemm<true, true, 32>(p, s) : dispatch_igemm<false, true, 32>(p, s);
    } else {
        rc = vec ? dispatch_igemm<true, false, 32>(p, s) : dispatch_igemm<false, false, 32>(p, s);
    }
    if (rc || p.ksplit <= 1) return rc;
    return launch_splitk_epilogue(p, s);
}

void stylex_wgrad_plan(const ConvKParams& p, int* tn, int* tc, int* splits, long* split_len) {
    // tile 128x128 when both channel counts are large, else 64x64
    bool big = p.N > 64 && p.Ck > 64;
    *tn = big ? 2 : 1;
    *tc = big ? 2 : 1;
    int bt = big ? 128 : 64;
    long tiles = (long)((p.N + bt - 1) / bt) * ((p.Ck + bt - 1) / bt) * p.KH * p.KW;
    long want = (1024 + tiles - 1) / tiles;                 // aim at ~1024 blocks (4 per CU)
    long max_by_len = ((long)p.M + 4 * BK - 1) / (4 * BK);  // at least 4 K-tiles per split
    long sp = want < 1 ? 1 : want;
    if (sp > max_by_len) sp = max_by_len;
    if (sp < 1) sp = 1;
    if (sp > 1024) sp = 1024;
    long len = (((long)p.M + sp - 1) / sp + BK - 1) / BK * BK;
    sp = ((long)p.M + len - 1) / len;
    *splits = (int)sp;
    *split_len = len;
}

int stylex_launch_wgrad(ConvKParams p, float* partial, float* dw_oihw, int precision, hipStream_t s, float* db, int* db_done) {
    if (db_done) *db_done = 0;
    if (precision == STYLEX_BF16 && !p.s2d_c && stylex_wgrad_pipe_applicable(p)) {  // (s2d: stylex_launch_wgrad_s2d_folded)
        int slices = 0, tps, blocks, bias_done = 0;
        stylex_wgrad_pipe_plan(p, &slices, &tps, &blocks);
        if (db) p.bias_partial = partial + (long)slices * p.N * 9 * p.Ck;  // behind the weight-gradient partials
        int rc = stylex_launch_wgrad_pipe(p, partial, s, &slices, &bias_done);
        if (rc != STYLEX_NOT_APPLICABLE) {
            if (rc) return rc;
            const bool with_db = db && bias_done;
            launch_wgrad_reduce(partial, dw_oihw, p.N, p.Ck, 9, slices, s, wg_out_of(p), with_db ? p.bias_partial : nullptr, with_db ? db : nullptr);
            if (with_db && db_done) *db_done = 1;
            return (int)hipGetLastError();
        }
        p.bias_partial = nullptr;
    }
    if (precision == STYLEX_BF16 && stylex_wgrad_tr_applicable(p)) {
        int ts = 0;
        int rc = stylex_launch_wgrad_tr(p, partial, s, &ts);
        if (rc) return rc;
        launch_wgrad_reduce(partial, dw_oihw, p.N, p.Ck, p.KH * p.KW, ts, s, wg_out_of(p));
        return (int)hipGetLastError();
    }
    int tn, tc, splits;
    long split_len;
    stylex_wgrad_plan(p, &tn, &tc, &splits, &split_len);
    p.split_len = split_len;
    p.y = partial;
    int bt = tn == 2 ? 128 : 64;
    int T = p.KH * p.KW;
    int blocks = ((p.N + bt - 1) / bt) * ((p.Ck + bt - 1) / bt) * T * splits;
    bool vec = (p.Ck % 4 == 0) && (p.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.a) & 15) == 0) &&
               ((reinterpret_cast<uintptr_t>(p.a2) & 15) == 0) &&
               (!p.a_scale || (reinterpret_cast<uintptr_t>(p.a_scale) & 15) == 0) &&
               (!p.a2_scale || (reinterpret_cast<uintptr_t>(p.a2_scale) & 15) == 0);
    int rc;
    bool bf = precision == STYLEX_BF16;
    if (tn == 2) {
        rc = bf ? (vec ? launch_wgrad<2, 2, true, true>(p, blocks, s) : launch_wgrad<2, 2, false, true>(p, blocks, s))
                : (vec ? launch_wgrad<2, 2, true, false>(p, blocks, s) : launch_wgrad<2, 2, false, false>(p, blocks, s));
    } else {
        rc = bf ? (vec ? launch_wgrad<1, 1, true, true>(p, blocks, s) : launch_wgrad<1, 1, false, true>(p, blocks, s))
                : (vec ? launch_wgrad<1, 1, true, false>(p, blocks, s) : launch_wgrad<1, 1, false, false>(p, blocks, s));
    }
    if (rc) return rc;
    launch_wgrad_reduce(partial, dw_oihw, p.N, p.Ck, T, splits, s, wg_out_of(p));
    return (int)hipGetLastError();
}

int stylex_launch_wgrad_s2d_folded(ConvKParams p, float* partial, float* dw_oihw, hipStream_t s) {
    int slices = 0;
    int rc = stylex_launch_wgrad_pipe(p, partial, s, &slices, nullptr);
    if (rc) return rc == STYLEX_NOT_APPLICABLE ? STYLEX_EINVAL : rc;
    launch_wgrad_reduce(partial, dw_oihw, p.N, p.s2d_c, 9, slices, s, wg_out_of(p));
    return (int)hipGetLastError();
}

int stylex_launch_pack(const float* w, void* wf, void* wb, int N, int C, int T, int dtype, hipStream_t s) {
    long total = (long)N * C * T;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (dtype == STYLEX_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, w, (unsigned short*)wf,
                           (unsigned short*)wb, N, C, T);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)wf, (float*)wb, N, C, T);
    return (int)hipGetLastError();
}

int stylex_launch_pack_s2d(const float* w, void* wf, void* wb, int N, int C, hipStream_t s) {
    long total = (long)N * C * 36;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_s2d_kernel, dim3(blocks), dim3(256), 0, s, w, (unsigned short*)wf, (unsigned short*)wb, N, C);
    return (int)hipGetLastError();
}

int stylex_launch_fold_s2d(const float* dw2, float* dw, int N, int C, hipStream_t s) {
    long total = (long)N * C * 9;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fold_weight_s2d_kernel, dim3(blocks), dim3(256), 0, s, dw2, dw, N, C);
    return (int)hipGetLastError();
}
