// Internal (non-ABI) declarations shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stylex_hip.h"
#include "act_io.h"

// GEMM-view of one convolution launch.  "source" = gathered activation tensor
// [B][Hi][Wi][Ck] (x for fprop/wgrad, dy for dgrad); "dest" = [B][Ho][Wo][N].
struct ConvKParams {
    const float* a;         // gathered activations
    const float* a_scale;   // [B][Ck] per-sample channel scale applied while staging (or null)
    const void* w;          // packed weights [N][T][Ck], fp32 (F32 mode) or bf16 (BF16 mode)
    const float* a2;        // second activation operand dy [M][N]   (wgrad)
    const float* a2_scale;  // [B][N]                                (wgrad)
    float* y;
    const float* bias;
    const float* out_scale;
    const float* noise;
    const float* noise_w;
    const float* noise_b;
    const float* residual;
    float res_scale;
    int noise_stride;
    int B, Hi, Wi, Ck, N, KH, KW, Ho, Wo, stride, pad;
    int transposed;   // 0: forward gather  ih = oh*s + kh - p ; 1: data-gradient gather ih = (oh + p - kh)/s
    int phase_major;  // rows ordered by (oh&1, ow&1) first (transposed stride-2 only)
    int act_bf16;     // activation tensors (a, a2, y, residual) are bf16 instead of fp32 (STYLEX_BF16_ACT)
    int s2d_c;        // >0: the conv is a 3x3/s2 conv in space-to-depth form: source channels = 4 sub-positions x s2d_c,
                      //     structurally-zero (tap, sub-position) pairs are skipped (conv_halo)
    int flip_taps;    // halo kernel: read weight tap 8-t for compute tap t (data gradient of a 3x3/s1/p1 conv)
    int M;            // B*Ho*Wo
    int flags;
    long split_len;   // wgrad: pixels per split (multiple of 32)
    int ksplit;       // igemm split-K: number of K slices (1 = none)
    int kt_per_split; // igemm split-K: K-tiles per slice
    float* partial;   // igemm split-K: [ksplit][B*Ho*Wo*N] raw accumulators
    unsigned char* mask;            // STYLEX_EPI_MASK_OUT: sign bits of the stored output, [M][N/8] bytes
    const unsigned char* gate_mask; // STYLEX_EPI_GATE_MASK: the activation gate of a data gradient as such a mask
    int dry;                        // launchers: run the applicability checks only, launch nothing (mask-support query)
    float* bias_partial;            // wgrad: [splits][N] per-split sums of dy over the pixels (bias gradient), or null
    // space-to-depth FORWARD only (conv_halo_dma.hip): a second K segment — the 1x1 residual conv of a DiscriminatorBlock,
    // y += x2[b, oh, ow, :] . w2[n, :] — accumulated behind the four sub-position phases of the 3x3 / stride-2 conv
    const void* x2;                 // [B][Ho][Wo][c2] bf16 (the block input at the even pixels), or null
    const void* w2;                 // [N][c2] bf16
    int c2;
    // weight gradient only: the reduce launch writes dw = (wg_accumulate ? dw : 0) + wg_scale * sum (wg_scale 0 = 1);
    // a bias sum produced by the same launch likewise
    float wg_scale;
    int wg_accumulate;
};

int stylex_launch_igemm(ConvKParams p, int precision, void* workspace, int64_t workspace_bytes, hipStream_t s);
int64_t stylex_igemm_workspace_bytes(const ConvKParams& p, int precision);
void stylex_wgrad_plan(const ConvKParams& p, int* tn, int* tc, int* splits, long* split_len);
// db != null: also produce db[n] = sum over pixels of dy[., n] when the selected kernel can (then *db_done = 1)
int stylex_launch_wgrad(ConvKParams p, float* partial, float* dw_oihw, int precision, hipStream_t s, float* db = nullptr,
                        int* db_done = nullptr);
int stylex_launch_pack(const float* w, void* wf, void* wb, int N, int C, int T, int dtype, hipStream_t s);

#define STYLEX_NOT_APPLICABLE (-100)

// Timing hook (stylex_timing_kernels): a launcher names the kernel it is about to launch, as rocprofv3 prints it
// (printf-style; thread-local, no allocation).  A timed C-ABI call attributes its hipEvent interval to the LAST name
// noted inside it (the main kernel of a multi-launch call notes itself last).
void stylex_note_kernel(const char* fmt, ...);

#ifdef __HIPCC__
// bit k = (bf16 element k of the 16-byte vector > 0), with the float comparison the tensor-gate path uses
__device__ __forceinline__ unsigned stylex_sign_bits8(uint4 v) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned m = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        m |= (__uint_as_float(w[d] << 16) > 0.f ? 1u : 0u) << (2 * d);
        m |= (__uint_as_float(w[d] & 0xffff0000u) > 0.f ? 1u : 0u) << (2 * d + 1);
    }
    return m;
}
#endif
// 3x3/s1/p1 bf16 kernel with the input halo resident in LDS (conv_halo.hip)
int stylex_launch_halo(const ConvKParams& p, hipStream_t s);

// LDS-DMA variant of the halo kernel for the >= 128-channel unscaled layers (conv_halo_dma.hip)
int stylex_launch_halo_dma(const ConvKParams& p, hipStream_t s);

// first layer of D / encoder: 3x3 over the padded RGB slot (C = 8) to 64 channels, forward (conv_rgb.hip)
int stylex_launch_rgb(const ConvKParams& p, hipStream_t s);

// persistent, software-pipelined LDS-DMA kernel for the unmodulated 3x3/s1 layers (conv_pipe.hip)
int stylex_launch_pipe(const ConvKParams& p, hipStream_t s);
int stylex_launch_line64(const ConvKParams& p, hipStream_t s);

// round 5: data gradient of the space-to-depth stride-2 conv, all four sub-positions per block (conv_s2d_dgrad.hip)
int stylex_launch_s2d_dgrad(const ConvKParams& p, hipStream_t s);
// round 5: forward of the space-to-depth stride-2 conv as one pipelined K loop over the sub-position phases (conv_s2d_fwd.hip)
int stylex_launch_s2d_fwd(const ConvKParams& p, hipStream_t s);

// LDS-DMA implicit GEMM for the <= 8x8 px layers (conv_gather.hip): writes fp32 partials and fills p.ksplit / p.partial
// for the split-K epilogue kernel
int stylex_launch_gather(ConvKParams& p, void* workspace, int64_t workspace_bytes, hipStream_t s);
int64_t stylex_gather_workspace_bytes(const ConvKParams& p);

// round 5: pipelined LDS-DMA weight gradient of the 3x3/s1/p1 bf16 layers with whole 64-channel tiles (conv_wgrad_pipe.hip)
bool stylex_wgrad_pipe_applicable(const ConvKParams& p);
void stylex_wgrad_pipe_plan(const ConvKParams& p, int* slices, int* tiles_per_split, int* blocks);
int stylex_launch_wgrad_pipe(ConvKParams p, float* partial, hipStream_t s, int* slices_out, int* bias_done = nullptr);
// space-to-depth stride-2 conv (p.s2d_c > 0): partial slices and result in the FOLDED layout [N][9][s2d_c] -> dw[N][s2d_c][3][3]
int stylex_launch_wgrad_s2d_folded(ConvKParams p, float* partial, float* dw_oihw, hipStream_t s);

// general bf16 weight gradient with LDS transpose reads (conv_wgrad_tr.hip)
bool stylex_wgrad_tr_applicable(const ConvKParams& p);
void stylex_wgrad_tr_plan(const ConvKParams& p, int* mode, int* splits, long* split_len);
int stylex_launch_wgrad_tr(ConvKParams p, float* partial, hipStream_t s, int* splits_out);

// Weight taps (bit kh*3+kw of the 3x3 frame) that are structurally non-zero for sub-position s = sy*2+sx of a
// stride-2 3x3/pad-1 convolution rewritten over the space-to-depth input (frame offset -1 -> kh2 = 0 needs
// the odd sub-row/col, offset 0 -> kh2 = 1 takes both, offset +1 never contributes).
__host__ __device__ inline unsigned stylex_s2d_tap_mask(int s) {
    const int sy = s >> 1, sx = s & 1;
    unsigned m = 0;
    for (int kh = 0; kh < 2; ++kh)
        for (int kw = 0; kw < 2; ++kw)
            if ((kh == 1 || sy == 1) && (kw == 1 || sx == 1)) m |= 1u << (kh * 3 + kw);
    return m;
}
int stylex_launch_pack_s2d(const float* w, void* wf, void* wb, int N, int C, hipStream_t s);
int stylex_launch_fold_s2d(const float* dw2, float* dw, int N, int C, hipStream_t s);
