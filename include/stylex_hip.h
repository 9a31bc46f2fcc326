/*
 * stylex_hip.h — C-ABI of libstylex_hip.so: hand-written gfx950 (MI355X / CDNA4)
 * kernels for the StylEx adversarial train step.
 *
 * The reference (NoahVl/Explaining-In-Style-Reproducibility-Study) has no native
 * layer of its own: every GPU kernel it runs comes from PyTorch/cuDNN through the
 * Python ops cited below.  This header is the boundary a maintainer would bind
 * (ctypes stub in INTEGRATION.md) to replace those ops on MI355X.
 *
 * Conventions:
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *    borrowed for the duration of the enqueue only;
 *  - activations are fp32 or bf16 (`act_dtype`) NHWC ("channels_last"):
 *    x[b][h][w][c]; weights arrive in the reference's OIHW parameter layout and
 *    are repacked by stylex_pack_weight();
 *  - every entry point only ENQUEUES work on `stream` (a hipStream_t passed as
 *    void*), never allocates/frees device memory, never synchronises;
 *  - return value: 0 on success, a hipError_t code (>0) from the runtime, or a
 *    negative STYLEX_E* code for bad arguments.  Never throws.
 *  - `shape` arrays are int64 host arrays; layout documented per function.
 *  - `precision`: arithmetic of the MFMA contraction —
 *      STYLEX_F32  v_mfma_f32_32x32x2_f32   (exact fp32, parity mode)
 *      STYLEX_BF16 v_mfma_f32_32x32x16_bf16 (operands rounded to bf16 when staged
 *                                            into LDS, fp32 accumulate), fp32 activation tensors.
 *      STYLEX_BF16_ACT  the same MFMA path with bf16 ACTIVATION tensors in HBM (every
 *                       pointer documented as "activation" is then bf16; scales, biases,
 *                       weight gradients and workspaces stay fp32).
 */
#ifndef STYLEX_HIP_H
#define STYLEX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STYLEX_F32 0
#define STYLEX_BF16 1
#define STYLEX_BF16_ACT 2 /* bf16 MFMA AND bf16 activation tensors (x, y, dy, dx, residual); halves HBM traffic */

#define STYLEX_EINVAL (-1)   /* bad shape / unsupported configuration */
#define STYLEX_EWORKSPACE (-2) /* workspace too small */

/* conv shape vector, shared by the three conv entry points:
 *   [0]=B [1]=Hi [2]=Wi [3]=C (input channels) [4]=N (output channels)
 *   [5]=KH [6]=KW [7]=stride [8]=pad [9]=Ho [10]=Wo                      */
#define STYLEX_CONV_NSHAPE 11

/* epilogue flags for stylex_conv2d_fwd */
#define STYLEX_EPI_BIAS 1      /* + bias[n]                                          */
#define STYLEX_EPI_LRELU 2     /* leaky_relu(., 0.2)                                 */
#define STYLEX_EPI_OSCALE 4    /* * out_scale[b][n]  (demodulation coefficient)      */
#define STYLEX_EPI_NOISE 8     /* + noise[b][w][h] * noise_w[n] + noise_b[n] (sic: transposed) */
#define STYLEX_EPI_RESIDUAL 16 /* (acc + residual[m][n]) * res_scale                 */
#define STYLEX_EPI_RELU 32     /* max(v, 0) last (instead of LRELU; frozen ResNet blocks) */
#define STYLEX_EPI_NOISE_NAT 128 /* with STYLEX_EPI_NOISE: the noise plane is in NATURAL order, value used at (h,w) is
                                 * noise[b][h][w] (the caller transposed the reference's plane once per generator
                                 * forward): the epilogue then reads 4 consecutive pixels with one 16-byte load */
#define STYLEX_EPI_MASK_OUT 256 /* fwd, with STYLEX_EPI_LRELU: also write epi->mask, one bit per stored output element
                                 * ([B][Ho][Wo][N/8] bytes; bit k of byte q of a pixel = (y[8q+k] > 0)) — everything a
                                 * later STYLEX_EPI_GATE_MASK / stylex_blur3x3_s2d_bwd_gate_mask needs of y, at 1/16 of its
                                 * bytes.  Only some kernels write it: ask stylex_conv_mask_supported() first.          */
#define STYLEX_EPI_GATE_MASK 512 /* bwd_data: as STYLEX_EPI_GATE with the gate given as such a bit mask in epi->mask       */
#define STYLEX_EPI_GATE 64     /* bwd_data only: dx *= (gate[m][c] > 0 ? 1 : res_scale) — the derivative of the
                                * (Leaky)ReLU that produced this conv's input, applied while dx is written; the gate
                                * tensor (= that input, [B][Hi][Wi][C], activation dtype) travels in epi->residual */

/* One-time per-process/per-device initialisation (kernel attributes). */
int stylex_init(int device);
const char* stylex_version(void);

/* Repack an OIHW parameter W[N][C][KH][KW] (fp32) into the two K-contiguous
 * layouts the implicit-GEMM kernels read:
 *   w_fwd[n][t][c]  (t = kh*KW+kw)   — forward / "fprop" operand
 *   w_bwd[c][t][n]                   — data-gradient operand
 * Either output may be NULL.  shape = {N, C, KH, KW}.  The packed element type follows `precision`:
 * fp32 for STYLEX_F32, bf16 (RNE) for STYLEX_BF16 — the conv entry points expect the matching one.
 * Replaces: the implicit weight handling inside F.conv2d
 * (reference stylex/stylex_train.py:660 and nn.Conv2d at :724-736, :771, :881). */
int stylex_pack_weight(const float* w_oihw, void* w_fwd, void* w_bwd, const int64_t* shape, int precision,
                       void* stream);

/* y[b,ho,wo,n] = epi( sum_{kh,kw,c} in_scale[b][c] * x[b,ho*s+kh-p,wo*s+kw-p,c] * w_fwd[n][kh*KW+kw][c] )
 * Replaces F.conv2d in Conv2DMod.forward (stylex_train.py:647-667, with the
 * modulation applied as in_scale = style+1 and demodulation as out_scale) and
 * nn.Conv2d in DiscriminatorBlock / Generator.initial_conv / final_conv
 * (:724-744, :771, :881) including the fused bias + LeakyReLU(0.2) (:340-341),
 * the noise add (:696-705) and the residual merge (:743).
 * Optional pointers may be NULL when the corresponding flag is clear. */
typedef struct {
    const float* in_scale;  /* [B][C]  or NULL */
    const float* bias;      /* [N] */
    const float* out_scale; /* [B][N] */
    const float* noise;     /* [B][S][S] image noise plane; value used at (h,w) is noise[b][w][h] */
    int64_t noise_stride;   /* S */
    const float* noise_w;   /* [N] */
    const float* noise_b;   /* [N] */
    const void* residual;   /* [B][Ho][Wo][N], activation dtype */
    float res_scale;
    int32_t s2d_c;          /* >0: space-to-depth form of a 3x3/stride-2/pad-1 conv (see stylex_blur3x3_s2d_fwd):
                             * `shape` describes the 3x3/s1/p1 conv over 4*s2d_c channels, the weights come from
                             * stylex_pack_weight_s2d, structurally-zero taps are skipped.  s2d_c % 64 == 0. */
    void* mask;             /* STYLEX_EPI_MASK_OUT: written; STYLEX_EPI_GATE_MASK: read (see the flags)          */
} stylex_conv_epilogue;

/* 1 when a launch with this shape / flags / precision (16-byte aligned tensors) runs on a kernel that writes
 * (which = 0: stylex_conv2d_fwd with STYLEX_EPI_MASK_OUT) or reads (which = 1: stylex_conv2d_bwd_data with
 * STYLEX_EPI_GATE_MASK) the activation bit mask; 0 otherwise — such a launch is then rejected with STYLEX_EINVAL. */
int stylex_conv_mask_supported(const int64_t* shape, int which, int flags, int precision);

/* Optional split-K workspace for launches that cannot fill the chip (few output tiles, long K):
 * query with stylex_conv2d_workspace_bytes(shape, which, precision) (which: 0 = fwd, 1 = bwd_data;
 * may return 0).  Passing workspace = NULL is always valid (no split, same result up to summation
 * order); with a workspace the slices are reduced in fixed order (deterministic). */
int64_t stylex_conv2d_workspace_bytes(const int64_t* shape, int which, int precision);

int stylex_conv2d_fwd(const void* x, const void* w_fwd, void* y, const int64_t* shape, int flags,
                      const stylex_conv_epilogue* epi, int precision, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* Tail of a DiscriminatorBlock in one launch (reference stylex_train.py:724, :733-743, bf16 activations):
 *   y = ( conv3x3_s2(blur(x))  [space-to-depth form: x_s2d {B,Ho,Wo,4C}, w_fwd_s2d from stylex_pack_weight_s2d, s2d_c = C]
 *       + conv1x1_s2(block input) [x_res {B,Ho,Wo,res_c} = the block input at the even pixels (stylex_subsample2_fwd),
 *                                  w_res {N,res_c} bf16, res_c % 8 == 0]
 *       + bias [N: the sum of the two convs' biases] ) * scale
 * The residual conv is a second K segment of the same accumulators (one more tap phase over x_res) instead of a GEMM
 * whose bf16 result the conv's epilogue reads back.  shape = the 11-entry conv shape of the 3x3 part
 * {B,Ho,Wo,4C,N,3,3,1,1,Ho,Wo}.  stylex_conv2d_s2d_res_supported() = 1 when a kernel takes the launch (16-byte aligned
 * tensors); stylex_conv2d_s2d_res_fwd returns STYLEX_EINVAL otherwise. */
int stylex_conv2d_s2d_res_supported(const int64_t* shape, int64_t s2d_c, int64_t res_c);
int stylex_conv2d_s2d_res_fwd(const void* x_s2d, const void* w_fwd_s2d, const void* x_res, const void* w_res, const float* bias,
                              void* y, const int64_t* shape, int64_t s2d_c, int64_t res_c, float scale, void* stream);

/* dx[b,hi,wi,c] = sum_{kh,kw,n} dy[b,(hi+p-kh)/s,(wi+p-kw)/s,n] * w_bwd[c][t][n]   (exact division only)
 * Replaces the input-gradient half of aten::convolution_backward issued by
 * autograd for the call sites above (and, through create_graph=True, the
 * double-backward chain of gradient_penalty, stylex_train.py:296-303). */
/* Modulated form: epi->in_scale [B][N] scales dy while it is staged (demodulation
 * coefficient) and, with STYLEX_EPI_OSCALE, epi->out_scale [B][C] scales dx
 * (style+1).  flags may only contain STYLEX_EPI_OSCALE and STYLEX_EPI_GATE (the activation-derivative pass of the
 * layer below fused into this launch: reference F.leaky_relu backward, stylex_train.py:724-731); epi may be NULL
 * when flags == 0. */
int stylex_conv2d_bwd_data(const void* dy, const void* w_bwd, void* dx, const int64_t* shape, int flags,
                           const stylex_conv_epilogue* epi, int precision, void* workspace, int64_t workspace_bytes,
                           void* stream);

/* dw[n][c][kh][kw] (OIHW, fp32) = sum_{b,ho,wo} dy[b,ho,wo,n] * x[b,ho*s+kh-p,wo*s+kw-p,c]
 * Deterministic two-stage split-K reduction through `workspace`.
 * Replaces the weight-gradient half of aten::convolution_backward. */
int64_t stylex_conv2d_bwd_weight_workspace_bytes(const int64_t* shape);
/* x_scale [B][C] / dy_scale [B][N] (either may be NULL) are the per-sample
 * modulation / demodulation factors of the modulated conv, applied while staging. */
int stylex_conv2d_bwd_weight(const void* x, const void* dy, float* dw, void* workspace, int64_t workspace_bytes,
                             const int64_t* shape, const float* x_scale, const float* dy_scale, int s2d_c,
                             int precision, void* stream);
/* As stylex_conv2d_bwd_weight; additionally db[n] = sum over (b, ho, wo) of dy[., n] — the bias gradient of the same
 * layer (reference: the bias half of aten::convolution_backward) — when the kernel that serves the shape can produce it
 * from the dy tiles it has staged anyway (one extra MFMA per k-step against a vector of ones; today the LDS-DMA kernel):
 * *db_written = 1 then, else 0 and db is untouched (the caller reduces dy itself). */
int stylex_conv2d_bwd_weight_bias(const void* x, const void* dy, float* dw, float* db, int* db_written, void* workspace,
                                  int64_t workspace_bytes, const int64_t* shape, const float* x_scale,
                                  const float* dy_scale, int s2d_c, int precision, void* stream);
/* Round 5: as stylex_conv2d_bwd_weight_bias (db may be NULL) with an output stage in the reduce launch:
 *   dw = (accumulate ? dw : 0) + out_scale * sum        (db likewise when it is written)
 * computed as two rounded fp32 operations, i.e. bit-identical to storing the sum, multiplying it in place and letting
 * autograd's AccumulateGrad add it to an existing gradient.  out_scale replaces the multi-tensor multiply of
 * DiscriminatorBlock's 1/sqrt(2) (reference stylex_train.py:743: (x + res) * (1 / math.sqrt(2))) on the weight gradients
 * computed from the unscaled output gradient; accumulate lets the second use of a twice-used parameter (the encoder of a
 * generator phase: E(x) and E(G(x)), stylex_train.py:1383-1395) add into the first use's gradient tensor instead of
 * handing the engine a second tensor to add. */
int stylex_conv2d_bwd_weight_ex(const void* x, const void* dy, float* dw, float* db, int* db_written, void* workspace,
                                int64_t workspace_bytes, const int64_t* shape, const float* x_scale, const float* dy_scale,
                                int s2d_c, float out_scale, int accumulate, int precision, void* stream);

/* Elementwise / resampling entry points take `act_dtype`: 0 = fp32 activations, 1 = bf16 activations
 * (the storage type of STYLEX_BF16_ACT); arithmetic is fp32 either way.
 *
 * Bilinear x2, align_corners=False (nn.Upsample, stylex_train.py:614,679) and its adjoint.
 * shape = {B, H, W, C} of the LOW-resolution tensor.  Index rule (exact):
 *   out[2k] = .25*in[max(k-1,0)] + .75*in[k];  out[2k+1] = .75*in[k] + .25*in[min(k+1,n-1)] */
int stylex_upsample2x_bilinear_fwd(const void* x, void* y, const int64_t* shape, int act_dtype, void* stream);
int stylex_upsample2x_bilinear_bwd(const void* dy, void* dx, const int64_t* shape, int act_dtype, void* stream);

/* 3x3 binomial blur /16 with reflect border (Blur.forward, stylex_train.py:144-153 ->
 * kornia.filters.filter2d(normalized=True, border_type='reflect')) and its adjoint.
 * shape = {B, H, W, C}.  Index rule (exact): -1 -> 1, H -> H-2. */
int stylex_blur3x3_reflect_fwd(const void* x, void* y, const int64_t* shape, int act_dtype, void* stream);
int stylex_blur3x3_reflect_bwd(const void* dy, void* dx, const int64_t* shape, int act_dtype, void* stream);

/* The RGB skip path of a GeneratorBlock in ONE pass (RGBBlock.forward, stylex_train.py:618-629: skip add :622-623,
 * then self.upsample = nn.Sequential(nn.Upsample(x2, bilinear), Blur()) :613-616, :625-626):
 *   out[B,2H,2W,C] = blur3x3_reflect(upsample2x_bilinear(rgb + prev)),   prev may be NULL (first block).
 * shape = {B, H, W, C} of the LOW-resolution tensors.  Both resamplers are separable and linear: the kernel applies
 * their composition as one 3x3 stencil over (rgb + prev) with exact border rules (clamp of the upsample, reflect of the
 * blur) and rounds once.  _bwd is the adjoint: dx[B,H,W,C] from dy[B,2H,2W,C] (the gradient of rgb AND of prev). */
int stylex_rgb_up_blur_add_fwd(const void* rgb, const void* prev, void* out, const int64_t* shape, int act_dtype,
                               void* stream);
int stylex_rgb_up_blur_add_bwd(const void* dy, void* dx, const int64_t* shape, int act_dtype, void* stream);

/* Space-to-depth pipeline for the blur + stride-2 conv of DiscriminatorBlock (stylex_train.py:733-742):
 * the blur writes its [B,H,W,C] result as [B,H/2,W/2,4C] (channel = ((h&1)*2+(w&1))*C + c), which turns the
 * 3x3/s2/p1 conv into a 3x3/s1/p1 conv over 4C channels whose weights (stylex_pack_weight_s2d, bf16,
 * from the OIHW parameter {N,C,3,3}) are zero for 27 of the 36 (tap, sub-position) pairs; the conv
 * kernels skip those (epi.s2d_c / s2d_c argument), so exactly the 9*C products of the original conv are
 * computed while every operand is read once.  stylex_fold_weight_grad_s2d maps the weight gradient of
 * the s2d conv ([N][4C][3][3]) back to the parameter layout.  H, W even. */
int stylex_blur3x3_s2d_fwd(const void* x, void* y_s2d, const int64_t* shape, int act_dtype, void* stream);
int stylex_blur3x3_s2d_bwd(const void* dy_s2d, void* dx, const int64_t* shape, int act_dtype, void* stream);
/* The two blur adjoints with the activation derivative of the layer below fused into the store:
 * dx = adjoint(dy) * (gate > 0 ? 1 : slope), gate = the blur's forward input (the LeakyReLU output of
 * DiscriminatorBlock.net, reference stylex_train.py:726-733), shape/dtype of dx. */
int stylex_blur3x3_reflect_bwd_gate(const void* dy, const void* gate, float slope, void* dx, const int64_t* shape,
                                    int act_dtype, void* stream);
int stylex_blur3x3_s2d_bwd_gate(const void* dy_s2d, const void* gate, float slope, void* dx, const int64_t* shape,
                                int act_dtype, void* stream);
/* The same with the gate given as the bit mask a forward conv wrote (STYLEX_EPI_MASK_OUT: [B][H][W][C/8] bytes): the
 * pass then reads 1/16 of the gate tensor's bytes.  bf16 activations, C % 8 == 0, H >= 8 (STYLEX_EINVAL otherwise). */
int stylex_blur3x3_s2d_bwd_gate_mask(const void* dy_s2d, const void* mask, float slope, void* dx, const int64_t* shape,
                                     int act_dtype, void* stream);
/* dst[b,2i,2j,:] += src[b,i,j,:] in place; shape = the FULL-resolution {B,H,W,C} of dst.  Sum of the two input
 * gradients of a DiscriminatorBlock (3x3 path + zero-inserted gradient of the 1x1/stride-2 conv_res path,
 * reference stylex_train.py:739-743) without the zero-inserted tensor. */
int stylex_add_at_even(const void* src, void* dst, const int64_t* shape, int act_dtype, void* stream);

/* Even-pixel gather y[b,i,j,:] = x[b,2i,2j,:] and its adjoint (zero insertion).  shape = the FULL-resolution
 * {B,H,W,C}; the low-resolution tensor is [B,(H+1)/2,(W+1)/2,C].  With these the reference's 1x1 / stride-2
 * residual convolution (DiscriminatorBlock.conv_res, stylex_train.py:724) runs as a contiguous 1x1 / stride-1
 * convolution of the gathered pixels. */
int stylex_subsample2_fwd(const void* x, void* y, const int64_t* shape, int act_dtype, void* stream);
int stylex_subsample2_bwd(const void* dy, void* dx, const int64_t* shape, int act_dtype, void* stream);
int stylex_pack_weight_s2d(const float* w_oihw, void* w_fwd, void* w_bwd, const int64_t* shape, void* stream);
int stylex_fold_weight_grad_s2d(const float* dw_s2d, float* dw_oihw, const int64_t* shape, void* stream);
/* Round 5: the weight gradient of that stride-2 conv in ONE call, already in the parameter layout dw[N][s2d_c][3][3]
 * (the nn.Conv2d(…, 3, padding=1, stride=2) weight of DiscriminatorBlock.downsample, stylex_train.py:733-736) —
 * x2 / dy / shape / workspace as stylex_conv2d_bwd_weight with s2d_c > 0 (shape describes the 3x3/s1/p1 conv over
 * 4*s2d_c channels; workspace of stylex_conv2d_bwd_weight_workspace_bytes(shape)).  Served by the pipelined LDS-DMA
 * weight-gradient kernel, whose blocks each multiply the 1, 2 or 4 live taps of one sub-position and write the folded
 * layout directly: no dW2 tensor, no fold launch.  _supported returns 1 when a launch with this shape (bf16 activations,
 * 16-byte aligned tensors) runs there, else 0 — then stylex_conv2d_bwd_weight_s2d returns STYLEX_EINVAL and the caller
 * uses stylex_conv2d_bwd_weight + stylex_fold_weight_grad_s2d. */
int stylex_conv2d_bwd_weight_s2d_supported(const int64_t* shape, int s2d_c, int precision);
/* out_scale / accumulate: the output stage of stylex_conv2d_bwd_weight_ex. */
int stylex_conv2d_bwd_weight_s2d(const void* x2, const void* dy, float* dw_oihw, void* workspace, int64_t workspace_bytes,
                                 const int64_t* shape, int s2d_c, float out_scale, int accumulate, int precision, void* stream);

/* y = leaky_relu(x + bias[c] (+ noise[b][w][h]*noise_w[c] + noise_b[c]), 0.2)
 * (nn.Conv2d bias + leaky_relu, stylex_train.py:340-341,726-731; noise add :696-714).
 * shape = {B, H, W, C}.  bias/noise pointers may be NULL.  bwd: dx = dy * (y>0 ? 1 : 0.2). */
int stylex_bias_act_fwd(const void* x, const float* bias, const float* noise, int64_t noise_stride,
                        const float* noise_w, const float* noise_b, void* y, const int64_t* shape, int act_dtype,
                        void* stream);
int stylex_bias_act_bwd(const void* dy, const void* y, void* dx, const int64_t* shape, int act_dtype, void* stream);

/* out[r] = sum_j x[r][j]^2   (per-sample squared L2 norm; gradient_penalty :302,
 * calc_pl_lengths :316).  shape = {rows, cols}.  One wavefront-shuffle + LDS
 * reduction per row block; deterministic. */
int stylex_rowwise_sumsq(const float* x, float* out, const int64_t* shape, void* stream);

/* RGB input of the first DiscriminatorBlock (reference stylex_train.py:724-726 reads a 3-channel image): any strided
 * [B][3][H][W] view (strides in ELEMENTS for b, c, h, w), fp32 or bf16 -> bf16 NHWC [B][H][W][8] with channels 3..7
 * zero — the 16-byte slot the vector load paths of the conv kernels want — in one pass. */
int stylex_pad_rgb8(const void* x, void* y, const int64_t* shape_bhw, const int64_t* strides_bchw, int x_is_bf16, void* stream);

/* ---- elementwise tail of the frozen classifier's conv layers ------------------------------------
 * The classifier (reference stylex/resnet_classifier.py:29-71, a torchvision ResNet-18 in eval mode) keeps its
 * convolutions on the stock library; an eval-mode BatchNorm is the affine map y = x * scale[c] + shift[c]
 * (scale = gamma / sqrt(var + eps), shift = beta - mean * scale), so BasicBlock's `bn -> relu` and
 * `bn -> (+ identity) -> relu` are one pass each, and the stem's `bn -> relu -> maxpool(3, 2, 1)` is one pass.
 * Dense fp32 NCHW tensors, [B][C][HW]; scale / shift [C].  `residual`, `gres`, `idx` may be NULL.
 *   fwd:  y = act(x * scale[c] + shift[c] + residual),  act = relu when `relu` != 0
 *   bwd:  gres = gy * [y > 0] (all of gy without relu),  gx = gres * scale[c]
 *   maxpool fwd: y[oh][ow] = max_{3x3 window, stride 2, pad 1} relu(x * scale[c] + shift[c]);  idx = window-local
 *        position (kh * 3 + kw) of the first strict maximum in row-major order (ATen's rule), 255 where the maximum
 *        is not positive;  Ho = (H - 1) / 2 + 1
 *   maxpool bwd: gx[h][w] = scale[c] * sum of gy over the windows whose idx names (h, w)   (a gather: deterministic) */
int stylex_affine_act_nchw_fwd(const float* x, const float* scale, const float* shift, const float* residual, float* y, int64_t B,
                               int64_t C, int64_t HW, int relu, void* stream);
int stylex_affine_act_nchw_bwd(const float* gy, const float* y, const float* scale, float* gx, float* gres, int64_t B, int64_t C,
                               int64_t HW, int relu, void* stream);
int stylex_affine_relu_maxpool_fwd(const float* x, const float* scale, const float* shift, float* y, unsigned char* idx, int64_t B,
                                   int64_t C, int64_t H, int64_t W, void* stream);
int stylex_affine_relu_maxpool_bwd(const float* gy, const unsigned char* idx, const float* scale, float* gx, int64_t B, int64_t C,
                                   int64_t H, int64_t W, void* stream);

/* ---- classifier input: bilinear resize + normalisation (round 6) --------------------------------------------------------
 * ResNet.classify_images (reference stylex/resnet_classifier.py:56-71): torchvision's tensor resize to 224 x 224
 * (== F.interpolate(mode='bilinear', align_corners=False), no antialias) followed by (x - mean[c]) / std[c], one pass.
 * sh = {B, C, Hi, Wi, Ho, Wo}; x fp32 read through its element strides {b, c, h, w} (any layout); y dense fp32 NCHW;
 * mean / stdv [C] or both NULL.  bwd: the exact adjoint (gy dense [B][C][Ho][Wo] -> gx dense [B][C][Hi][Wi], divided by
 * stdv[c] when given), a gather in fixed order.  Index rule: src = in / out * (dst + 0.5) - 0.5 clamped at 0,
 * i0 = (int)src, i1 = i0 + (i0 < in - 1), weights (1 - frac, frac). */
int stylex_resize_norm_fwd(const float* x, float* y, const float* mean, const float* stdv, const int64_t* sh, const int64_t* strides,
                           void* stream);
int stylex_resize_norm_bwd(const float* gy, float* gx, const float* stdv, const int64_t* sh, void* stream);

/* ---- layout bridges: the library's dense fp32 NCHW tensors <-> this library's bf16 NHWC tensors (round 6) -------------
 * Where a frozen network keeps part of its path on the library's fp32 convolutions (the classifier forward, north_star; the
 * LPIPS stem) and the rest runs on the bf16 kernels: frozen_resnet._ResNetBodyHybrid, lpips_alex._taps_bf16.  C % 8 == 0.
 *   stylex_nchw_f32_to_nhwc_bf16:  y[b][p][c] = bf16(relu ? max(x[b][c][p], 0) : x[b][c][p])
 *   stylex_nhwc_bf16_to_nchw_f32:  gx[b][c][p] = float(g[b][p][c]) * (gate == NULL || gate[b][p][c] > 0)   (gate: bf16 NHWC) */
/* out = (y > 0) ? a + b : 0, bf16 tensors of `numel` elements in the same (any) layout, numel % 8 == 0, b may be NULL: the ReLU
 * gate of a residual block's output applied to the sum of the gradients of its two consumers (frozen_resnet._ResNetBodyHybrid). */
int stylex_relu_gate_add(const void* a, const void* b, const void* y, void* out, int64_t numel, void* stream);
int stylex_nchw_f32_to_nhwc_bf16(const float* x, void* y, int64_t B, int64_t C, int64_t HW, int relu, void* stream);
int stylex_nhwc_bf16_to_nchw_f32(const void* g, const void* gate, float* gx, int64_t B, int64_t C, int64_t HW, void* stream);

/* ---- LPIPS-AlexNet's max-pools on bf16 NHWC feature maps (round 6) --------------------------------
 * nn.MaxPool2d(kernel_size=3, stride=2) of torchvision's AlexNet features[2] / [5] (LPIPS-AlexNet: reference
 * stylex/stylex_train.py:404 through lpips 0.1.4) for this library's bf16 channels_last taps (lpips_alex._taps_bf16).
 *   sh = {B, Hi, Wi, C}, C % 8 == 0, Hi, Wi >= 3;  Ho = (Hi - 3) / 2 + 1, Wo likewise.
 *   fwd: y[b][oh][ow][c] = max over the window; idx[b][oh][ow][c] (one byte) = kh * 3 + kw of the FIRST maximum in scan order
 *        (a NaN wins) — ATen's rule, so the gradient goes where max_pool2d_with_indices_backward sends it.
 *   bwd: gx[b][ih][iw][c] = sum of gy over the (at most 2 x 2) windows that chose (ih, iw); fp32 sum, one rounding. */
int stylex_maxpool3s2_nhwc_fwd(const void* x, void* y, void* idx, const int64_t* sh, void* stream);
int stylex_maxpool3s2_nhwc_bwd(const void* gy, const void* idx, void* gx, const int64_t* sh, void* stream);

/* ---- input gradient of a frozen network's first convolution (round 6) ---------------------------
 * The K x K / stride-S stem over the 3-channel image of the frozen classifier (torchvision ResNet conv1: 7 x 7 / 2 / pad 3,
 * reference stylex/resnet_classifier.py:19, 56-71) and of LPIPS-AlexNet (11 x 11 / 4 / pad 2, reference stylex_train.py:404):
 *   dx[b][c][ih][iw] = sum_n sum_{kh,kw} dy[b][n][(ih + pad - kh) / S][(iw + pad - kw) / S] * w[n][c][kh][kw]
 * (terms whose divisions are exact and land inside dy) — the gradient torch.nn.grad.conv2d_input defines.  Dense fp32
 * NCHW: dy [B][N][Ho][Wo], w [N][C][K][K], dx [B][C][Hi][Wi]; sh = {B, N, Ho, Wo, C, K, S, pad, Hi, Wi}; C <= 4,
 * K <= 15, S in {1, 2, 4}.  Replaces the library's dense transposed convolution in the backward of the two stems
 * (frozen_resnet.py / lpips_alex.py `_FirstConv`); fixed summation order. */
int stylex_conv_image_grad(const float* dy, const float* w, float* dx, const int64_t* sh, void* stream);

/* ---- LPIPS distance of one feature tap -----------------------------------------------------------
 * reconstruction_loss (reference stylex_train.py:404-438) calls lpips.LPIPS(net='alex') (lpips 0.1.4): per tap
 *   n = f / (sqrt(sum_c f^2) + 1e-10),  d[b][p] = sum_c lin[c] * (n0 - n1)^2,  out[b] = mean_p d[b][p].
 * f0, f1: dense fp32 [B][C][HW]; lin [C].  fwd writes per-block sums of d / HW to partial[b * partial_stride + block],
 * block = 0 .. ceil(HW / 64) - 1 (the caller adds them up, over all taps at once, in fixed order) and the per-pixel
 * norms sqrt(sum_c f^2) to r0 / r1 [B][HW] (NULL = not kept).  bwd: g0 / g1 (either may be NULL) = gout[b] * d out[b] / d f,
 * the chain rule through both normalisations; an all-zero pixel yields NaN exactly like the reference's sqrt backward. */
int stylex_lpips_tap_fwd(const float* f0, const float* f1, const float* lin, float* partial, float* r0, float* r1, int64_t B,
                         int64_t C, int64_t HW, int64_t partial_stride, void* stream);
int stylex_lpips_tap_bwd(const float* f0, const float* f1, const float* lin, const float* r0, const float* r1, const float* gout,
                         float* g0, float* g1, int64_t B, int64_t C, int64_t HW, void* stream);

/* The same tap on bf16 NHWC features f[b][p][c] (round 6: LPIPS-AlexNet on this library's bf16 convolution kernels in the speed
 * mode, stylex/lpips_alex.py): C % 8 == 0, 16-byte aligned pointers; per-block sums of d / HW over blocks of 32 pixels,
 * block = 0 .. ceil(HW / 32) - 1; norms r0 / r1 fp32 [B][HW]; bwd writes g0 / g1 as bf16 NHWC. */
int stylex_lpips_tap_nhwc_fwd(const void* f0, const void* f1, const float* lin, float* partial, float* r0, float* r1, int64_t B,
                              int64_t C, int64_t HW, int64_t partial_stride, void* stream);
int stylex_lpips_tap_nhwc_bwd(const void* f0, const void* f1, const float* lin, const float* r0, const float* r1, const float* gout,
                              void* g0, void* g1, int64_t B, int64_t C, int64_t HW, void* stream);

/* ---- modulated-conv coefficients (SURVEY §8(b) `demod_coeff` / `bwd_style`) --------------------
 * Conv2DMod.forward (reference stylex_train.py:650-656) in the batched form:
 *   s1[b][i] = style[b][i] + 1,   d[b][o] = rsqrt( sum_i s1[b][i]^2 * wsq[o][i] + eps ),
 *   wsq[o][i] = sum_k W[o][i][k]^2  (k over the KH*KW taps).
 * All tensors fp32 and dense: style, s1 [B][C]; d, gd [B][O]; wsq [O][C]; w, gw [O][C][K].
 * stylex_weight_sumsq: one launch per weight VERSION (cache it until the optimiser step).
 * stylex_modcoeff_fwd: writes s1 and d.
 * stylex_modcoeff_bwd: first-order backward through d (gd = dL/dd):
 *   gstyle[b][i] = (gs1 ? gs1[b][i] : 0) + 2 s1[b][i] * sum_o dq[b][o] wsq[o][i],  dq = -gd d^3 / 2
 *   gw[o][i][k]  = 2 w[o][i][k] * sum_b dq[b][o] s1[b][i]^2
 * gs1 = the gradient that reaches s1 directly (from the conv's input scale), or NULL; gstyle / gw may be NULL
 * (that gradient is not needed).  Fixed summation order (deterministic). */
int stylex_weight_sumsq(const float* w, float* wsq, int64_t O, int64_t C, int64_t K, void* stream);
int stylex_modcoeff_fwd(const float* style, const float* wsq, float* s1, float* d, int64_t B, int64_t C, int64_t O,
                        float eps, void* stream);
int stylex_modcoeff_bwd(const float* gd, const float* d, const float* s1, const float* wsq, const float* w,
                        const float* gs1, float* gstyle, float* gw, int64_t B, int64_t C, int64_t O, int64_t K,
                        void* stream);

/* ---- fused fast path (steps that need no double backward) ------------------------------------
 * Backward-side kernels that fold the LeakyReLU mask, a scale and the per-(image, channel)
 * reductions into ONE pass.  Tensors NHWC fp32, shape = {B, H, W, C}, C % 4 == 0, C <= 1024.
 * Each launch writes partial[b][chunk][k][C] with nchunks = stylex_reduce_chunks(shape); the caller
 * sums over chunks (and over b where the parameter is per channel).  Deterministic.
 *
 * stylex_act_bwd_reduce:   dx = dy * scale * (lrelu ? (y>0 ? 1 : slope) : 1), slope = .2 (lrelu==1) or 0 (lrelu==2, ReLU);
 *                          partial = sum_pixels dx
 *   -> backward of lrelu(conv + bias) (stylex_train.py:726-731) and of (x+res)/sqrt(2) (:743);
 *      dx may be NULL (reduction only).
 * stylex_modconv_bwd_prep: for y = lrelu(d*z + noise[b,w,h]*nw[c] + nb[c]) (:700-714):
 *      gz = gy * lrelu'(y);  partial[.][0] = sum gz*(d*z), [1] = sum gz*noise, [2] = sum gz
 * stylex_scale_reduce:     gx = t * s[b][c];  partial = sum_pixels x*t   (gradient wrt style+1, :650) */
int stylex_reduce_chunks(const int64_t* shape);
int stylex_act_bwd_reduce(const void* dy, const void* y, void* dx, float* partial, const int64_t* shape, int nchunks,
                          int lrelu, float scale, int act_dtype, void* stream);
int stylex_modconv_bwd_prep(const void* gy, const void* y, const float* noise, int64_t noise_stride,
                            const float* noise_w, const float* noise_b, void* gz, float* partial,
                            const int64_t* shape, int nchunks, int lrelu, int act_dtype, void* stream);
/* Same, and the STORED gradient is gz * gz_scale[b][c] (gz_scale = the demodulation coefficient d [B][C]): both consumers
 * of gz (data gradient and weight gradient of the modulated conv) want gz * d, so their operands become scale-free
 * and the data gradient can take the LDS-DMA kernels.  The three sums are those of the unscaled gz. */
int stylex_modconv_bwd_prep_scaled(const void* gy, const void* y, const float* noise, int64_t noise_stride,
                                   const float* noise_w, const float* noise_b, const float* gz_scale, void* gz,
                                   float* partial, const int64_t* shape, int nchunks, int lrelu, int act_dtype,
                                   void* stream);
int stylex_scale_reduce(const void* x, const void* t, const float* s, void* gx, float* partial, const int64_t* shape,
                        int nchunks, int act_dtype, void* stream);

/* ---- to-RGB layer (RGBBlock.forward, stylex_train.py:618-621: Conv2DMod(C, 3, kernel 1, demod=False)) ----------
 * Streaming kernels for the 3-output-channel modulated 1x1 convolution; bf16 NHWC activations only,
 * C a power of two in [8, 512] (else STYLEX_NOT_APPLICABLE).  shape = {B, H, W, C}.
 *   s1 = style + 1, fp32 [B][C];  w = the layer's weight, fp32 [3][C] (OIHW with 1x1 taps).
 * stylex_torgb_fwd:  y[b,p,n] = sum_c x[b,p,c] * s1[b,c] * w[n,c]   -> y bf16 [B][H][W][4], channel 3 = 0.
 * stylex_torgb_bwd:  gy bf16 [B][H][W][4] (channel 3 ignored);
 *                    gx[b,p,c] = s1[b,c] * sum_n gy[b,p,n] w[n,c]   (gx may be NULL);
 *                    partial[b][chunk][n][c] = sum over the chunk's pixels of x[b,p,c] * gy[b,p,n],
 *                    chunk < stylex_torgb_chunks(shape); the caller sums over chunks and forms
 *                    d style = sum_n w*T  and  dW = sum_b s1*T.  Deterministic. */
int stylex_torgb_chunks(const int64_t* shape);
int stylex_torgb_fwd(const void* x, const float* s1, const float* w, void* y, const int64_t* shape, void* stream);
int stylex_torgb_bwd(const void* x, const void* gy, const float* s1, const float* w, void* gx, float* partial,
                     const int64_t* shape, void* stream);

/* Per-kernel timing hook: when enabled every conv launch is bracketed
 * by hipEvents on its stream; stylex_timing_report returns, per kernel class
 * (0=fwd,1=bwd_data,2=bwd_weight): launches, total ms, total algorithmic FLOPs and total algorithmic HBM
 * bytes (activations in + out once at their storage width, weights once). */
int stylex_timing_enable(int on);
/* on != 0: the CALLING THREAD's launches are not recorded until the matching stylex_timing_pause(0) (nestable) — the frozen
 * networks' layers that run on these kernels (bf16 LPIPS-AlexNet, the classifier's data gradient) stay out of the StylEx
 * conv classes, as SURVEY §8(d) prescribes; bench.py reports them under `frozen_nets`. */
int stylex_timing_pause(int on);
int stylex_timing_report(int kernel_class, int64_t* launches, double* total_ms, double* total_flops,
                         double* total_bytes);
/* Per-layer view of the same measurements, one row per (class, conv shape): meta[r][10] = {class, B, Hi, Wi, C, N, KH,
 * stride, s2d_c, launches}, vals[r][3] = {total ms, total algorithmic FLOPs, total algorithmic bytes}.  Returns the
 * number of rows written (<= cap).  bench.py builds the per-layer mixed (HBM / MFMA) roofline from it. */
int stylex_timing_layers(int64_t* meta, double* vals, int64_t cap);
/* Per-kernel view: one row per (class, name of the kernel the call launched, spelled as rocprofv3 prints it — e.g.
 * "conv3x3_pipe_kernel<128, 0>").  names = cap slots of 112 bytes, meta[r][2] = {class, launches}, vals[r][3] as above.
 * bench.py names the dominant kernel of its roofline record from it, so that the record can be checked against the
 * rocprofv3 --kernel-trace --stats summary under profiles/. */
int stylex_timing_kernels(char* names, int64_t* meta, double* vals, int64_t cap);

/* ---- Adam step + operand copies in one launch (bf16 speed mode) -------------------------------------------------
 * Reference: Adam(lr, betas=(0.5, 0.9)) of StylEx.G_opt / D_opt (stylex_train.py:957-959), stepped at :1357 / :1449.
 * One launch applies the Adam update (rule of torch._fused_adam_, no weight decay / amsgrad) to a LIST of fp32 tensors
 * and rewrites, from the updated values, the derived copies the conv kernels read: kind PACK = the bf16 operand layouts
 * of stylex_pack_weight (a = [N][T][C], b = [C][T][N]; either may be NULL; a 1x1 weight's [N][C] GEMM matrix is the
 * same layout with T = 1), PACK_S2D = those of stylex_pack_weight_s2d, SUMSQ = stylex_weight_sumsq (a = fp32 [N][C]);
 * every copy holds scale * w.  `step` points to the tensor's step counter (a device float, ALREADY incremented).
 * N = 0 marks a flat tensor (numel elements, no copies).  descs / block_map live in device memory: block_map[i] = index
 * of the tensor that block i works on, blocks of one tensor consecutive from first_block, their number given by
 * stylex_adam_pack_tensor_blocks() (a host helper; -1 for an unsupported geometry: T > 9 or N*C*T != numel). */
#define STYLEX_ADAM_COPY_PACK 0
#define STYLEX_ADAM_COPY_PACK_S2D 1
#define STYLEX_ADAM_COPY_SUMSQ 2
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    const float* step;
    int64_t numel;
    int32_t N, C, T, nvar;
    int64_t first_block;
    double lr, beta1, beta2, eps;
    struct {
        int32_t kind;
        float scale;
        void* a;
        void* b;
    } var[4];
} stylex_adam_tensor;
int64_t stylex_adam_pack_tensor_blocks(int64_t numel, int32_t N, int32_t C, int32_t T);
int stylex_adam_pack_step(const stylex_adam_tensor* descs_dev, const int32_t* block_map_dev, int64_t n_blocks, void* stream);

/* K10 — the scalar loss reductions of the train step, one forward and one backward launch each (fp32; deterministic:
 * fixed reduction trees, no atomics).  Results and incoming gradients are DEVICE scalars: no host synchronisation.
 *   hinge       mode 0: out = mean(relu(1 + real) + relu(1 - fake))   (hinge_loss, stylex_train.py:386-387)
 *               mode 1: out = mean(fake)                               (gen_hinge_loss, :382-383; real ignored)
 *   pl_lengths  len[b] = sqrt(mean_l(sum_d g[b][l][d]^2)), shape = {B, L, D}   (calc_pl_lengths :316)
 *   kl_logits   out = sum_b sum_k p_real (log p_real - log p_fake) / B over log-softmaxed logit rows, shape = {B, K}
 *               (classifier_kl_loss :421-438 with KLDivLoss(reduction='batchmean', log_target=True) :406)
 *   l1_mean     out = mean(|a - b|) over n elements (nn.L1Loss :404-405, used at :415-418).  shape4 = NULL: both
 *               operands in one linear element order.  Otherwise the logical tensor is the 4-D index space {shape4}
 *               (n = its product < 2^32) walked with the last index fastest; an operand with a NULL stride array is
 *               linear in that order, the other is addressed through its element strides.  Operand dtypes 0 = fp32 /
 *               1 = bf16; `partial` = stylex_l1_mean_chunks(n) floats of workspace; a gradient takes the dtype and the
 *               addressing of its operand
 * bwd: gradient pointers may be NULL where a gradient is not wanted (at least one must be given). */
int stylex_hinge_fwd(const float* real, const float* fake, float* out, int64_t n, int mode, void* stream);
int stylex_hinge_bwd(const float* real, const float* fake, const float* gout, float* greal, float* gfake, int64_t n, int mode,
                     void* stream);
int stylex_pl_lengths_fwd(const float* g, float* len, const int64_t* shape, void* stream);
int stylex_pl_lengths_bwd(const float* g, const float* len, const float* glen, float* gg, const int64_t* shape, void* stream);
int stylex_kl_logits_fwd(const float* real, const float* fake, float* out, const int64_t* shape, void* stream);
int stylex_kl_logits_bwd(const float* real, const float* fake, const float* gout, float* greal, float* gfake, const int64_t* shape,
                         void* stream);
int64_t stylex_l1_mean_chunks(int64_t n);
int stylex_l1_mean_fwd(const void* a, const void* b, float* partial, float* out, int64_t n, int a_dtype, int b_dtype,
                       const int64_t* shape4, const int64_t* a_strides4, const int64_t* b_strides4, void* stream);
int stylex_l1_mean_bwd(const void* a, const void* b, const float* gout, void* ga, void* gb, int64_t n, int a_dtype, int b_dtype,
                       const int64_t* shape4, const int64_t* a_strides4, const int64_t* b_strides4, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STYLEX_HIP_H */
