"""Differentiable ops of the StylEx hot path on the HIP kernels.

Every op is a ``torch.autograd.Function`` whose backward is itself built from
Functions of this file, so arbitrary-order differentiation works — needed for
``gradient_penalty`` (double backward through D, reference
stylex/stylex_train.py:296-303) and ``calc_pl_lengths`` (through G, :306-316).
The three convolution roles are closed under differentiation:

    Conv(x,w)    --bwd-->  Dgrad(gy,w), Wgrad(x,gy)
    Dgrad(gy,w)  --bwd-->  Conv(ggx,w), Wgrad(ggx,gy)
    Wgrad(x,gy)  --bwd-->  Dgrad(gy,ggw), Conv(x,ggw)

Tensors are logically NCHW (reference API) and physically NHWC
(``torch.channels_last``); activations are fp32 in the parity mode and bf16 in
the speed mode (``set_precision``).  First-order-only passes take the fused
once-differentiable Functions (``set_fast``); double-differentiable passes use
the same fused forward kernels with a backward composed of the Functions above.

The module-level functions (``conv2d`` …) dispatch to an *implementation
object*.  The default and only product implementation is the HIP one below; it
raises if the library or a GPU is missing.  ``tests/`` may install a CPU test
double with ``use_impl`` to exercise the host logic without a GPU.
"""
import math

import os

import torch
from contextlib import nullcontext as _nullcontext
import torch.nn.functional as F

import hip_backend as hb

_PRECISION = hb.F32


_NAMES = {"fp32": hb.F32, "f32": hb.F32, "bf16": hb.BF16_ACT, "bf16_f32act": hb.BF16}


def set_precision(name):
    """'fp32'        exact f32 MFMA, fp32 tensors (parity mode);
    'bf16'        bf16 MFMA operands + bf16 activation tensors in HBM, fp32 accumulate / parameters / reductions;
    'bf16_f32act' bf16 MFMA operands with fp32 activation tensors."""
    global _PRECISION
    _PRECISION = _NAMES[name]


def get_precision():
    return {hb.F32: "fp32", hb.BF16_ACT: "bf16", hb.BF16: "bf16_f32act"}[_PRECISION]


def act_dtype():
    return hb.act_dtype(_PRECISION)


def _cl(t):
    """channels_last + the activation dtype of the current precision mode."""
    return hb.to_cl(t, hb.act_dtype(_PRECISION))


def _act(t):
    """Differentiable cast of an incoming tensor to the activation dtype (no-op in fp32 modes)."""
    if t is None:
        return None
    adt = hb.act_dtype(_PRECISION)
    return t if t.dtype == adt else t.to(adt)


# ------------------------------------------------------------------------------------------
# convolution triad
# ------------------------------------------------------------------------------------------


_INPUTS_ONLY = False


class inputs_only:
    """Context for `torch.autograd.grad(..., inputs=<activations>)` calls (gradient penalty, path length):
    autograd asks every Function for ALL its input gradients even when only the data path is wanted, so the
    weight/bias gradients of that first backward would be computed and thrown away.  Inside this context the
    conv Functions skip them."""

    def __enter__(self):
        global _INPUTS_ONLY
        self.prev, _INPUTS_ONLY = _INPUTS_ONLY, True

    def __exit__(self, *exc):
        global _INPUTS_ONLY
        _INPUTS_ONLY = self.prev


def _want_param_grad(ctx, i):
    return ctx.needs_input_grad[i] and not _INPUTS_ONLY


def _gemm_1x1(t, w, stride, pad, s2d):
    """A 1x1 / stride-1 conv in the bf16 mode is a plain GEMM: forward and data gradient of the composable
    (double-differentiable) triad go to hipBLASLt like the fused DiscriminatorBlock's residual conv (17-130 TF/s on the
    generic kernel, 3.2 ms per gradient-penalty step).  The WEIGHT gradient stays on the deterministic split-K kernel:
    letting ATen differentiate an addmm instead made two identically seeded Trainers diverge (its K = B*H*W weight-
    gradient GEMM takes a split-K algorithm with atomic accumulation; tools/determinism_check.py 4 of 4 runs)."""
    return (not s2d and stride == 1 and pad == 0 and _PRECISION == hb.BF16_ACT and t.is_cuda and t.dtype == torch.bfloat16
            and w.dim() == 4 and tuple(w.shape[2:]) == (1, 1) and os.environ.get("STYLEX_RES_GEMM", "1") != "0"
            and os.environ.get("STYLEX_RES_GEMM_DD", "1") != "0")


def _fwd(x, w, stride, pad, s2d, **epi):
    """conv forward; s2d = C of the original stride-2 conv when x is its space-to-depth image."""
    if _gemm_1x1(x, w, stride, pad, s2d) and not any(v is not None and v is not False for k, v in epi.items()
                                                      if k not in ("bias", "res_scale")):
        return hb.conv1x1_gemm_fwd(_cl(x), w, epi.get("bias"))
    if s2d:
        wf2, _ = hb.pack_weight_s2d(w)
        return hb.conv2d_fwd(x, None, 1, 1, _PRECISION, packed=wf2, w_shape=(w.shape[0], 4 * s2d, 3, 3), s2d_c=s2d, **epi)
    return hb.conv2d_fwd(x, w, stride, pad, _PRECISION, **epi)


def _bwd_data(gy, w, x_shape, stride, pad, s2d):
    if _gemm_1x1(gy, w, stride, pad, s2d):
        return hb.conv1x1_gemm_bwd_data(_cl(gy), w)
    if s2d:
        _, wb2 = hb.pack_weight_s2d(w)
        return hb.conv2d_bwd_data(gy, None, x_shape, 1, 1, _PRECISION, packed=wb2, w_shape=(w.shape[0], 4 * s2d, 3, 3),
                                  s2d_c=s2d)
    return hb.conv2d_bwd_data(gy, w, x_shape, stride, pad, _PRECISION)


def _bwd_weight(x, gy, w_shape, stride, pad, s2d):
    if s2d:
        return hb.conv2d_bwd_weight_s2d(x, gy, tuple(w_shape), _PRECISION)
    return hb.conv2d_bwd_weight(x, gy, w_shape, stride, pad, _PRECISION)


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, s2d=0):
        x = _cl(x)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, s2d)
        return _fwd(x, w, stride, pad, s2d)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad, s2d = ctx.cfg
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _Dgrad.apply(gy, w, tuple(x.shape), stride, pad, s2d)
        if _want_param_grad(ctx, 1):
            gw = _Wgrad.apply(x, gy, tuple(w.shape), stride, pad, s2d)
        return gx, gw, None, None, None


class _Dgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, w, x_shape, stride, pad, s2d=0):
        gy = _cl(gy)
        ctx.save_for_backward(gy, w)
        ctx.cfg = (x_shape, stride, pad, s2d)
        return _bwd_data(gy, w, x_shape, stride, pad, s2d)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        x_shape, stride, pad, s2d = ctx.cfg
        d_gy = d_w = None
        if ctx.needs_input_grad[0]:
            d_gy = _Conv.apply(ggx, w, stride, pad, s2d)
        if _want_param_grad(ctx, 1):
            d_w = _Wgrad.apply(ggx, gy, tuple(w.shape), stride, pad, s2d)
        return d_gy, d_w, None, None, None, None


class _Wgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gy, w_shape, stride, pad, s2d=0):
        x, gy = _cl(x), _cl(gy)
        ctx.save_for_backward(x, gy)
        ctx.cfg = (w_shape, stride, pad, s2d)
        return _bwd_weight(x, gy, w_shape, stride, pad, s2d)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        w_shape, stride, pad, s2d = ctx.cfg
        d_x = d_gy = None
        ggw = ggw.contiguous()
        if ctx.needs_input_grad[0]:
            d_x = _Dgrad.apply(gy, ggw, tuple(x.shape), stride, pad, s2d)
        if ctx.needs_input_grad[1]:
            d_gy = _Conv.apply(x, ggw, stride, pad, s2d)
        return d_x, d_gy, None, None, None, None


# ------------------------------------------------------------------------------------------
# bias + LeakyReLU(0.2)
# ------------------------------------------------------------------------------------------


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        x = _cl(x)
        y = hb.bias_act_fwd(x, bias)
        ctx.save_for_backward(y)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gx = _BiasActBwd.apply(gy, y)
        gb = None
        if ctx.has_bias and _want_param_grad(ctx, 1):
            gb = _channel_sum(gx) if (not torch.is_grad_enabled() and gx.is_cuda and hb.is_cl(gx)) else gx.sum(dim=(0, 2, 3), dtype=torch.float32)
        return gx, gb


class _BiasActBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, y):
        gy = _cl(gy)
        ctx.save_for_backward(y)
        return hb.bias_act_bwd(gy, y)

    @staticmethod
    def backward(ctx, ggx):
        (y,) = ctx.saved_tensors
        return _BiasActBwd.apply(ggx, y), None


# ------------------------------------------------------------------------------------------
# resampling (both linear: the adjoint's adjoint is the op itself)
# ------------------------------------------------------------------------------------------


class _Up(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return hb.upsample2x_fwd(_cl(x))

    @staticmethod
    def backward(ctx, gy):
        return _UpBwd.apply(gy)


class _UpBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy):
        return hb.upsample2x_bwd(_cl(gy))

    @staticmethod
    def backward(ctx, ggx):
        return _Up.apply(ggx)


class _Blur(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return hb.blur3x3_fwd(_cl(x))

    @staticmethod
    def backward(ctx, gy):
        return _BlurBwd.apply(gy)


class _BlurBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy):
        return hb.blur3x3_bwd(_cl(gy))

    @staticmethod
    def backward(ctx, ggx):
        return _Blur.apply(ggx)


class _Subsample2(torch.autograd.Function):
    """x[:, :, ::2, ::2] (dense); adjoint = zero insertion; the pair is closed under differentiation."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = tuple(x.shape[2:])
        return hb.subsample2_fwd(_cl(x))

    @staticmethod
    def backward(ctx, gy):
        return _ZeroInsert2.apply(gy, ctx.hw)


class _ZeroInsert2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, hw):
        return hb.subsample2_bwd(_cl(gy), hw)

    @staticmethod
    def backward(ctx, ggx):
        return _Subsample2.apply(ggx), None


class _RowSumSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d):
        ctx.save_for_backward(x2d)
        return hb.rowwise_sumsq(x2d)

    @staticmethod
    def backward(ctx, g):
        (x2d,) = ctx.saved_tensors
        return 2.0 * x2d * g[:, None]


# ------------------------------------------------------------------------------------------
# K10: the scalar losses of the step as one forward + one backward launch each (csrc/losses.hip).  First-order nodes:
# none of these losses is differentiated twice (the gradient penalty and the path-length term differentiate D / G
# twice, not the reduction on top — the backward of `pl_lengths` only FEEDS the double-backward graph of G).
# STYLEX_FUSED_LOSSES=0 keeps the torch compositions (A/B, bisecting).
# ------------------------------------------------------------------------------------------

_FUSED_LOSSES = os.environ.get("STYLEX_FUSED_LOSSES", "1") != "0"


def _loss_fusable(*ts):
    return (_FUSED_LOSSES and _IMPL is HipOps and
            all(t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.numel() > 0 for t in ts))


class _Hinge(torch.autograd.Function):
    """mode 0: mean(relu(1 + real) + relu(1 - fake)) (reference hinge_loss :386-387); mode 1: fake.mean() (:382-383)."""

    @staticmethod
    def forward(ctx, real, fake, mode):
        real = hb._f32c(real) if real is not None else None
        fake = hb._f32c(fake)
        ctx.save_for_backward(real, fake)
        ctx.mode = mode
        return hb.hinge_fwd(real, fake, mode)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        real, fake = ctx.saved_tensors
        greal, gfake = hb.hinge_bwd(real, fake, hb._f32c(g), ctx.mode == 0 and ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                    ctx.mode)
        return greal, gfake, None


class _PLLengths(torch.autograd.Function):
    """sqrt(mean_l(sum_d g^2)) per sample (reference calc_pl_lengths :316)."""

    @staticmethod
    def forward(ctx, g):
        g = hb._f32c(g)
        lengths = hb.pl_lengths_fwd(g)
        ctx.save_for_backward(g, lengths)
        return lengths

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, glen):
        g, lengths = ctx.saved_tensors
        return hb.pl_lengths_bwd(g, lengths, hb._f32c(glen))


class _KLLogits(torch.autograd.Function):
    """KLDivLoss(batchmean, log_target=True)(log_softmax(fake), log_softmax(real)) (reference :421-438)."""

    @staticmethod
    def forward(ctx, real, fake):
        real, fake = hb._f32c(real), hb._f32c(fake)
        ctx.save_for_backward(real, fake)
        return hb.kl_logits_fwd(real, fake)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        real, fake = ctx.saved_tensors
        return hb.kl_logits_bwd(real, fake, hb._f32c(g), ctx.needs_input_grad[0], ctx.needs_input_grad[1])


class _L1Mean(torch.autograd.Function):
    """nn.L1Loss() (reference :404-405): mean |a - b|."""

    @staticmethod
    def forward(ctx, a, b, walk):
        ctx.save_for_backward(a, b)
        ctx.walk = walk
        return hb.l1_mean_fwd(a, b, walk)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga, gb = hb.l1_mean_bwd(a, b, hb._f32c(g), ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.walk)
        return ga, gb, None


def hinge_loss(real, fake):
    if _loss_fusable(real, fake) and real.shape == fake.shape:
        return _Hinge.apply(real, fake, 0)
    return (F.relu(1 + real) + F.relu(1 - fake)).mean()


def gen_hinge_loss(fake):
    if _loss_fusable(fake):
        return _Hinge.apply(None, fake, 1)
    return fake.mean()


def pl_lengths(pl_grads):
    if _loss_fusable(pl_grads) and pl_grads.dim() == 3:
        return _PLLengths.apply(pl_grads)
    return (pl_grads ** 2).sum(dim=2).mean(dim=1).sqrt()


def kl_logits(real_logits, fake_logits):
    if _loss_fusable(real_logits, fake_logits) and real_logits.dim() == 2 and real_logits.shape == fake_logits.shape:
        return _KLLogits.apply(real_logits, fake_logits)
    return F.kl_div(F.log_softmax(fake_logits, dim=1), F.log_softmax(real_logits, dim=1), reduction="batchmean", log_target=True)


def l1_mean(a, b):
    if _loss_fusable(a, b):
        walk = hb.l1_walk(a, b)
        if walk is not None:
            return _L1Mean.apply(a, b, walk)
    return F.l1_loss(a, b)


# ------------------------------------------------------------------------------------------
# fused fast path: ONE forward kernel per conv (scales, bias, noise, residual merge, LeakyReLU in the
# epilogue) and fused backward bookkeeping.  Backward is NOT differentiable again, so it is used
# only when no double backward can be requested: under no_grad, or when the Trainer has declared
# the phase free of gradient-penalty / path-length terms (set_fast).
# ------------------------------------------------------------------------------------------

_FAST = False


def set_fast(flag):
    """Trainer hook: True when the coming forward/backward needs first-order gradients only."""
    global _FAST
    prev = _FAST
    _FAST = bool(flag)
    return prev


def fast_enabled():
    return _FAST or not torch.is_grad_enabled()


def _reducible(c):
    return c % 4 == 0 and c <= 1024


class _ConvBiasActFast(torch.autograd.Function):
    """y = lrelu?( (conv(x, w) + bias + residual) * res_scale )"""

    @staticmethod
    def forward(ctx, x, w, bias, residual, stride, pad, lrelu, res_scale):
        x = _cl(x)
        if residual is not None:
            residual = _cl(residual)
        y = hb.conv2d_fwd(x, w, stride, pad, _PRECISION, bias=bias, lrelu=lrelu, residual=residual,
                          res_scale=res_scale)
        ctx.save_for_backward(x, w, y if lrelu else None)
        ctx.cfg = (stride, pad, lrelu, float(res_scale) if residual is not None else 1.0, bias is not None,
                   residual is not None)
        ctx.untimed = hb.timing_paused()  # built inside hb.timing_pause() (a frozen network's layer): so is its backward
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, lrelu, scale, has_bias, has_res = ctx.cfg
        gy = _cl(gy)
        gb = None
        want_gb = has_bias and ctx.needs_input_grad[2]  # False in the generator phase (D frozen): no reduction launch
        if _reducible(gy.shape[1]):
            want_dx = lrelu or scale != 1.0
            if want_dx or want_gb:
                gz, gb = hb.act_bwd_reduce(gy, y, lrelu, scale, want_dx=want_dx, want_sum=want_gb)
                if not want_dx:
                    gz = gy
            else:
                gz = gy
        else:
            gz = hb.bias_act_bwd(gy, y) if lrelu else gy
            if scale != 1.0:
                gz = gz * scale
            gb = gz.sum(dim=(0, 2, 3), dtype=torch.float32) if want_gb else None
        with (hb.timing_pause() if ctx.untimed else _nullcontext()):
            gx = hb.conv2d_bwd_data(gz, w, tuple(x.shape), stride, pad, _PRECISION) if ctx.needs_input_grad[0] else None
            gw = hb.conv2d_bwd_weight(x, gz, tuple(w.shape), stride, pad, _PRECISION) if ctx.needs_input_grad[1] else None
        if not ctx.needs_input_grad[2]:
            gb = None
        return gx, gw, gb, (gz if has_res and ctx.needs_input_grad[3] else None), None, None, None, None


class _ConvBiasActDD(torch.autograd.Function):
    """Same fused forward kernel as _ConvBiasActFast; the backward is composed of differentiable Functions
    (activation-derivative kernel, the conv triad), so it can be differentiated again: this is what the
    real branch of a gradient-penalty step runs.  `s2d` = C when x is the space-to-depth image of a stride-2
    conv's input (see _DownS2DFast)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, stride, pad, lrelu, res_scale, s2d=0):
        x = _cl(x)
        if residual is not None:
            residual = _cl(residual)
        y = _fwd(x, w, stride, pad, s2d, bias=bias, lrelu=lrelu, residual=residual, res_scale=res_scale)
        ctx.save_for_backward(x, w, y if lrelu else None)
        ctx.cfg = (stride, pad, lrelu, float(res_scale) if residual is not None else 1.0, bias is not None,
                   residual is not None, s2d)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, lrelu, scale, has_bias, has_res, s2d = ctx.cfg
        gz = _BiasActBwd.apply(gy, y.detach()) if lrelu else gy
        if scale != 1.0:
            gz = gz * scale
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _Dgrad.apply(gz, w, tuple(x.shape), stride, pad, s2d)
        if _want_param_grad(ctx, 1):
            gw = _Wgrad.apply(x, gz, tuple(w.shape), stride, pad, s2d)
        if has_bias and _want_param_grad(ctx, 2):
            # a backward that is not itself recorded (the ordinary loss.backward() of a penalty step) takes the read-only
            # reduction kernel: 2.4x faster than the ATen reduce on the 256^2 tensors (.118 vs .285 ms at B=64)
            gb = _channel_sum(gz) if (not torch.is_grad_enabled() and gz.is_cuda and hb.is_cl(gz)) else gz.sum(dim=(0, 2, 3), dtype=torch.float32)
        return gx, gw, gb, (gz if has_res and ctx.needs_input_grad[3] else None), None, None, None, None, None


_NAT_NOISE = {}
import atexit as _atexit  # noqa: E402

_atexit.register(_NAT_NOISE.clear)  # device tensors are released before the interpreter / HIP runtime shut down


def _natural_noise(inoise):
    """The generator's noise plane in natural order, nat[b][h][w] = inoise[b][w][h][0] (the reference permutes the
    projected noise (0,3,2,1), :696-698, i.e. reads the plane transposed).  Transposed ONCE per generator forward (all
    14 noise layers of a forward share the plane, each cropping its top-left h x w corner) so that the conv epilogues
    read 4 consecutive pixels with one 16-byte load instead of 16 strided 4-byte loads: measured on the modulated convs
    with noise, 64->32 @256^2 .60 -> .31 ms and 32->32 @256^2 .49 -> .19 ms per launch (tools/bench_modconv.py)."""
    # one entry, keyed by the tensor OBJECT (held alive by the entry, so its address cannot be recycled by another
    # noise tensor) and its version counter (static graph input buffers are refilled in place)
    hit = _NAT_NOISE.get("k")
    if (hit is not None and hit[0] is inoise and hit[1] == inoise._version
            and not (inoise.is_cuda and torch.cuda.is_current_stream_capturing())):
        return hit[2]
    nat = inoise[:, :, :, 0].transpose(1, 2).contiguous().float()
    _NAT_NOISE["k"] = (inoise, inoise._version, nat)
    return nat


class _ModConvFast(torch.autograd.Function):
    """y = lrelu?( d[b,o] * conv(x * s1[b,i], w) + noise[b,w,h]*nw[o] + nb[o] )   (Conv2DMod + noise + act)"""

    @staticmethod
    def forward(ctx, x, s1, d, w, noise, nw, nb, pad, lrelu, noise_nat=None):
        x = _cl(x)
        # <= 8x8 px layers: the modulation is applied to the (tiny) input tensor up front, so that the launch can take
        # the LDS-DMA implicit-GEMM kernel (conv_gather.hip: a DMA cannot scale in flight); same rounding as the
        # in-kernel staging (product rounded to bf16 once)
        pre = _PRECISION == hb.BF16_ACT and s1 is not None and x.shape[2] * x.shape[3] <= 64 and x.shape[1] % 64 == 0
        ctx.pre = pre
        if pre:
            x_in, s_in = (x.float() * s1[:, :, None, None]).to(x.dtype).contiguous(memory_format=torch.channels_last), None
        else:
            x_in, s_in = x, s1
        if noise_nat is not None and noise_nat.shape[1] % 4 == 0:
            y = hb.conv2d_fwd(x_in, w, 1, pad, _PRECISION, in_scale=s_in, out_scale=d, noise=noise_nat, noise_w=nw,
                              noise_b=nb, lrelu=lrelu, noise_natural=True)
        else:
            y = hb.conv2d_fwd(x_in, w, 1, pad, _PRECISION, in_scale=s_in, out_scale=d, noise=noise, noise_w=nw, noise_b=nb,
                              lrelu=lrelu)
        ctx.save_for_backward(x, s1, d, w, noise, nw, nb, y if (lrelu or d is not None or noise is not None) else None)
        ctx.cfg = (pad, lrelu)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, s1, d, w, noise, nw, nb, y = ctx.saved_tensors
        pad, lrelu = ctx.cfg
        gy = _cl(gy)
        gd = gnw = gnb = None
        folded = False  # gz already carries the demodulation coefficient d
        if y is not None and _reducible(gy.shape[1]):
            fold = d is not None and _PRECISION != hb.F32 and os.environ.get("STYLEX_FOLD_D", "1") != "0"
            gz, sums = hb.modconv_bwd_prep(gy, y, noise, nw, nb, lrelu, gz_scale=d if fold else None)
            folded = fold
            if d is not None:
                gd = sums[:, 0] / d
            if noise is not None:
                gnw, gnb = sums[:, 1:3].sum(0)  # one reduction launch for both (the same per-element sums)
        elif y is not None:
            raise hb.StylexHipError("fused modulated conv needs C_out % 4 == 0")
        else:
            gz = gy
        gx = gs1 = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if folded:  # scale-free operand: eligible for the LDS-DMA kernels (halo / small-spatial gather)
                t = hb.conv2d_bwd_data(gz, w, tuple(x.shape), 1, pad, _PRECISION)
            elif ctx.pre and d is not None:  # small layer: demodulation applied to the (tiny) gradient tensor up front
                gzd = (gz.float() * d[:, :, None, None]).to(gz.dtype).contiguous(memory_format=torch.channels_last)
                t = hb.conv2d_bwd_data(gzd, w, tuple(x.shape), 1, pad, _PRECISION)
            else:
                t = hb.conv2d_bwd_data(gz, w, tuple(x.shape), 1, pad, _PRECISION, in_scale=d)
            if _reducible(x.shape[1]):
                gx, gs1 = hb.scale_reduce(x, t, s1, want_gx=ctx.needs_input_grad[0])
            else:
                gs1 = (x.float() * t.float()).sum(dim=(2, 3))
                gx = (t.float() * s1[:, :, None, None]).to(t.dtype)
        if ctx.needs_input_grad[3]:
            gw = hb.conv2d_bwd_weight(x, gz, tuple(w.shape), 1, pad, _PRECISION, x_scale=s1, dy_scale=None if folded else d)
        return gx, gs1, gd, gw, None, gnw, gnb, None, None, None


class _ToRGBFast(torch.autograd.Function):
    """RGBBlock's Conv2DMod(C, 3, 1, demod=False) (reference :611, :621) as one streaming kernel each way.
    Returns the 4-channel storage (channel 3 zero); the caller slices [:, :3]."""

    @staticmethod
    def forward(ctx, x, s1, w):
        x = _cl(x)
        ctx.save_for_backward(x, s1, w)
        return hb.torgb_fwd(x, s1, w)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, s1, w = ctx.saved_tensors
        gx, t = hb.torgb_bwd(x, _cl(gy), s1, w, want_gx=ctx.needs_input_grad[0])
        gs1 = gw = None
        if ctx.needs_input_grad[1]:
            gs1 = (t * w.reshape(1, 3, -1).float()).sum(dim=1)
        if ctx.needs_input_grad[2]:
            gw = (t * s1.float()[:, None, :]).sum(dim=0).reshape(w.shape).to(w.dtype)
        return gx, gs1, gw


class _RGBTailFast(torch.autograd.Function):
    """blur(upsample2x(rgb + prev)) of RGBBlock.forward (reference :622-626) as one kernel each way (4-channel NHWC
    storage of the RGB chain; once-differentiable: the composable chain serves steps that differentiate twice)."""

    @staticmethod
    def forward(ctx, rgb, prev):
        return hb.rgb_up_blur_add_fwd(_cl(rgb), _cl(prev) if prev is not None else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        g = hb.rgb_up_blur_add_bwd(_cl(gy))
        return g, (g if ctx.needs_input_grad[1] else None)


class _BlurS2D(torch.autograd.Function):
    """blur3x3 whose output is stored space-to-depth ([B,4C,H/2,W/2]) for the stride-2 conv that follows."""

    @staticmethod
    def forward(ctx, x):
        return hb.blur3x3_s2d_fwd(_cl(x))

    @staticmethod
    def backward(ctx, gy2):
        return _BlurS2DBwd.apply(gy2)


class _BlurS2DBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy2):
        return hb.blur3x3_s2d_bwd(_cl(gy2))

    @staticmethod
    def backward(ctx, ggx):
        return _BlurS2D.apply(ggx)


class _DownS2DFast(torch.autograd.Function):
    """(conv3x3_s2(x) + bias + residual) * res_scale with x given space-to-depth: runs as a 3x3/s1 conv over
    4C channels on the LDS-halo kernels, structurally-zero taps skipped (exactly the original 9*C work)."""

    @staticmethod
    def forward(ctx, x2, w, bias, residual, res_scale):
        x2 = _cl(x2)
        if residual is not None:
            residual = _cl(residual)
        n, c = w.shape[0], w.shape[1]
        wf2, _ = hb.pack_weight_s2d(w)
        y = hb.conv2d_fwd(x2, None, 1, 1, _PRECISION, bias=bias, residual=residual, res_scale=res_scale, packed=wf2,
                          w_shape=(n, 4 * c, 3, 3), s2d_c=c)
        ctx.save_for_backward(x2, w)
        ctx.cfg = (float(res_scale) if residual is not None else 1.0, bias is not None, residual is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        scale, has_bias, has_res = ctx.cfg
        n, c = w.shape[0], w.shape[1]
        gy = _cl(gy)
        want_dx = scale != 1.0
        want_gb = has_bias and ctx.needs_input_grad[2]
        gz, gsum = gy, None
        if want_dx or want_gb:
            gz, gsum = hb.act_bwd_reduce(gy, None, False, scale, want_dx=want_dx, want_sum=want_gb)
            if not want_dx:
                gz = gy
        gx2 = gw = None
        if ctx.needs_input_grad[0]:
            _, wb2 = hb.pack_weight_s2d(w)
            gx2 = hb.conv2d_bwd_data(gz, None, tuple(x2.shape), 1, 1, _PRECISION, packed=wb2, w_shape=(n, 4 * c, 3, 3),
                                     s2d_c=c)
        if ctx.needs_input_grad[1]:
            gw = hb.conv2d_bwd_weight_s2d(x2, gz, tuple(w.shape), _PRECISION)
        gb = gsum if want_gb else None
        return gx2, gw, gb, (gz if has_res and ctx.needs_input_grad[3] else None), None


def _fork_side(t):
    """Companion stream for the 1x1 residual path of a fused DiscriminatorBlock (networks._side_stream), or None."""
    # off by default: measured 699 vs 700 images/s with / without it (the fused block leaves too little idle time for
    # a second stream to fill); STYLEX_DBLOCK_SIDE=1 switches it on for experiments
    if os.environ.get("STYLEX_DBLOCK_SIDE", "0") != "1":
        return None
    import networks

    return networks._side_stream(t)


class _DBlockFast(torch.autograd.Function):
    """A whole DiscriminatorBlock (reference :721-744) as ONE autograd node on the fused kernels:

        res = conv1x1_s2(x) + b_r ;  y1 = lrelu(conv3x3(x) + b1) ;  y2 = lrelu(conv3x3(y1) + b2)
        out = (conv3x3_s2(blur(y2)) + b3 + res) / sqrt(2)              (last block: (y2 + res) / sqrt(2))

    Why one node: the backward can then fuse ACROSS the layers — the LeakyReLU derivative of y1 rides the store of
    conv2's data gradient and that of y2 the store of the blur adjoint (STYLEX_EPI_GATE: no separate
    activation-derivative pass over the two largest tensors of the block), the bias gradient of the residual conv and
    of the down conv is the same reduction, and the gradient of the 1x1/stride-2 path is added into the 3x3 path's
    input gradient at the even pixels in place (no zero-inserted tensor, no full-resolution add).  First-order only
    (used when no double backward can be requested, see set_fast)."""

    @staticmethod
    def forward(ctx, x, w_res, b_res, w1, b1, w2, b2, w3, b3, downsample):
        c = 1 / math.sqrt(2)
        cin = x.shape[1]
        if (cin == 3 and _PRECISION == hb.BF16_ACT and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)
                and os.environ.get("STYLEX_PAD_RGB", "1") != "0"):
            # RGB input, bf16 mode: cast + channels_last + zero-pad to one 16-byte channel slot in ONE pass
            x = hb.pad_rgb8(x.detach())
            # padded weights and their operand packs: cached per parameter version (they used to be rebuilt — zeros,
            # cat, pack, cast — in every forward and backward of block 0: ~50 launches per train() call)
            w1p, wrp = hb.pad_in_channels(w1, 5), hb.pad_in_channels(w_res, 5)
        elif cin == 3:  # RGB input: pad to one 16-byte channel slot (see _pad_rgb); gradients are sliced back
            x = _cl(_act(x))
            x, w1p, _ = _pad_rgb(x, w1)
            x = _cl(x)
            wrp = torch.cat([w_res, w_res.new_zeros(w_res.shape[0], x.shape[1] - 3, 1, 1)], dim=1)
        else:
            x = _cl(_act(x))
            w1p, wrp = w1, w_res
        # the 1x1 residual path (even-pixel gather + small GEMM) is independent of the two 3x3 convs until the merge:
        # companion HIP stream, joined before the kernel that merges
        side = _fork_side(x)
        # the 1x1 conv of the residual path is a plain GEMM: hipBLASLt in the bf16 mode (2-3x the generic kernel)
        res_gemm = _PRECISION == hb.BF16_ACT and os.environ.get("STYLEX_RES_GEMM", "1") != "0"
        conv_res = (lambda t: hb.conv1x1_gemm_fwd(t, wrp if cin == 3 else w_res, b_res,
                                                  owner=w_res if cin == 3 else None)) if res_gemm else (
            lambda t: hb.conv2d_fwd(t, wrp, 1, 0, _PRECISION, bias=b_res))
        # Blocks whose stride-2 conv runs on the LDS-DMA space-to-depth kernel take the residual conv as a second K segment
        # of that launch (hb.conv2d_s2d_res_fwd): no GEMM, no bf16 `res` tensor written and read back (round 4;
        # STYLEX_RES_FOLD=0 = the separate GEMM)
        n2_, h2_, w2_ = w2.shape[0], x.shape[2], x.shape[3]
        fold_res = (bool(downsample) and _PRECISION == hb.BF16_ACT and res_gemm and h2_ % 2 == 0 and w2_ % 2 == 0
                    and n2_ % 64 == 0 and x.shape[1] % 8 == 0 and os.environ.get("STYLEX_RES_FOLD", "1") != "0"
                    and hb.s2d_res_supported((x.shape[0], 4 * n2_, h2_ // 2, w2_ // 2), w_res.shape[0], n2_, x.shape[1]))
        res = None
        if fold_res:
            xs = hb.subsample2_fwd(x)
            side = None
        elif side is None:
            xs = hb.subsample2_fwd(x) if downsample else x
            res = conv_res(xs)
        else:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                xs = hb.subsample2_fwd(x) if downsample else x
                res = conv_res(xs)
            x.record_stream(side)
        # Activation bit masks: the backward needs y2 ONLY for its sign (the LeakyReLU derivative fused into the blur
        # adjoint) and y1 for its sign plus as the weight-gradient operand; where the forward kernel can write the sign
        # bits alongside (1/16 of the bytes) the gated backward passes read those instead of the tensors, and y2 is not
        # kept for the backward at all.  STYLEX_GATE_MASK=0: gate on the tensors.
        n2, h2, wd2 = w2.shape[0], x.shape[2], x.shape[3]
        s2d_next = bool(downsample) and (_PRECISION != hb.F32 and n2 % 64 == 0 and h2 % 2 == 0 and wd2 % 2 == 0
                                         and wd2 // 2 >= 16 and h2 // 2 >= 16)
        # (measured per launch at B=64, tools/bench_masks.py: gated data gradient 64->64 @256^2 .534 -> .445 ms, gated blur
        # adjoint .323 -> .273, 128 @128^2 .356 -> .327 / .159 -> .141; writing the mask costs the forward 0 - .012 ms;
        # at 64^2 and below the saving no longer covers that, so the masks are used from 128^2 up)
        use_m = (_PRECISION == hb.BF16_ACT and os.environ.get("STYLEX_GATE_MASK", "1") != "0"
                 and h2 * wd2 >= int(os.environ.get("STYLEX_GATE_MASK_MIN_PIXELS", 128 * 128)))
        pk1 = dict(packed=hb.pack_weight(w1p, True, False, _PRECISION, owner=w1)[0], w_shape=tuple(w1p.shape)) \
            if (cin == 3 and w1p is not w1 and isinstance(w1, torch.nn.Parameter)) else {}
        y1, m1 = hb.conv2d_fwd(x, w1p, 1, 1, _PRECISION, bias=b1, lrelu=True, want_mask=True, **pk1) if use_m else (
            hb.conv2d_fwd(x, w1p, 1, 1, _PRECISION, bias=b1, lrelu=True, **pk1), None)
        want_m2 = use_m and s2d_next and hb.blur_mask_ok((x.shape[0], n2, h2, wd2), x.dtype)
        y2, m2 = hb.conv2d_fwd(y1, w2, 1, 1, _PRECISION, bias=b2, lrelu=True, want_mask=True) if want_m2 else (
            hb.conv2d_fwd(y1, w2, 1, 1, _PRECISION, bias=b2, lrelu=True), None)
        s2d, xb = False, None
        def join():
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                res.record_stream(torch.cuda.current_stream())
                xs.record_stream(torch.cuda.current_stream())

        if downsample:
            n, h, w = y2.shape[1], y2.shape[2], y2.shape[3]
            s2d = (_PRECISION != hb.F32 and n % 64 == 0 and h % 2 == 0 and w % 2 == 0 and w // 2 >= 16 and h // 2 >= 16)
            if s2d:
                xb = hb.blur3x3_s2d_fwd(y2)
                join()
                wf2, _ = hb.pack_weight_s2d(w3)
                if fold_res:
                    wm = hb._bf16_matrix(wrp if cin == 3 else w_res, owner=w_res if cin == 3 else None)
                    bsum = hb.cached_vector("sum32", lambda a, b: a.float() + b.float(), b3, b_res)
                    out = hb.conv2d_s2d_res_fwd(xb, wf2, xs, wm, bsum, w3.shape[0], n, c)
                else:
                    out = hb.conv2d_fwd(xb, None, 1, 1, _PRECISION, bias=b3, residual=res, res_scale=c, packed=wf2,
                                        w_shape=(w3.shape[0], 4 * n, 3, 3), s2d_c=n)
            else:
                xb = hb.blur3x3_fwd(y2)
                join()
                out = hb.conv2d_fwd(xb, w3, 2, 1, _PRECISION, bias=b3, residual=res, res_scale=c)
        else:
            join()
            out = (y2 + res) * c
        assert m2 is None or s2d
        ctx.save_for_backward(x, xs if downsample else None, y1, y2 if m2 is None else None, xb, w_res, w1, w2, w3, m1, m2)
        ctx.y2_shape = tuple(y2.shape)
        ctx.cfg = (bool(downsample), s2d, cin, c)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        x, xs, y1, y2, xb, w_res, w1, w2, w3, m1, m2 = ctx.saved_tensors
        downsample, s2d, cin, c = ctx.cfg
        if xs is None:
            xs = x
        need = ctx.needs_input_grad
        want_x = need[0]
        want_b = need[2] or need[4] or need[6] or need[8]  # bias gradients (False for the frozen D of the G phase)
        want_w = need[1] or need[3] or need[5] or need[7]
        g_out = _cl(g_out)
        prec = _PRECISION
        gw_res = gb_res = gw1 = gb1 = gw2 = gb2 = gw3 = gb3 = gx = None
        # (. + res) / sqrt(2).  In the bf16 modes the constant c never touches the activation gradient: it is folded
        # into the packed data-gradient operands (c*w3, c*w_res) and into the (small) weight / bias gradients, which
        # removes one full pass over the block's output gradient; the fp32 parity mode scales explicitly.
        alg = downsample and prec != hb.F32
        if alg:
            gz3 = g_out
            gsum3 = _channel_sum(g_out, c) if want_b else None  # with its 1/sqrt(2)
        elif _reducible(g_out.shape[1]):
            gz3, gsum3 = hb.act_bwd_reduce(g_out, None, False, c, want_dx=True, want_sum=want_b)
        else:
            gz3 = g_out * c
            gsum3 = gz3.sum(dim=(0, 2, 3), dtype=torch.float32) if want_b else None
        wsc = c if alg else 1.0  # factor owed by the gradients computed from the unscaled gz3: the reduce launches' out_scale
        gb_res = gsum3  # the per-channel sum is the bias gradient of BOTH conv_res and the down conv
        if cin == 3:  # x was saved padded; only the weights need padding again
            extra = x.shape[1] - 3
            w1p, wrp = hb.pad_in_channels(w1, extra), hb.pad_in_channels(w_res, extra)
        else:
            w1p, wrp = w1, w_res
        # residual path (1x1 weight gradient + data gradient on the quarter-size tensor): companion stream, joined
        # before its results are used at the end
        side_bwd, side_out, main = _fork_side(g_out), None, torch.cuda.current_stream() if g_out.is_cuda else None
        res_gemm = prec == hb.BF16_ACT and os.environ.get("STYLEX_RES_GEMM", "1") != "0"

        def res_dgrad():
            if res_gemm:  # [M, N] x [N, C] on hipBLASLt, the 1/sqrt(2) folded into the bf16 weight copy as below
                return hb.conv1x1_gemm_bwd_data(gz3, wrp if cin == 3 else w_res, scale=c if alg else None,
                                                owner=w_res if cin == 3 else None)
            wbr = hb.pack_weight(wrp, False, True, prec, scale=c)[1] if alg else None
            return hb.conv2d_bwd_data(gz3, wrp, tuple(xs.shape), 1, 0, prec, packed=wbr, w_shape=tuple(wrp.shape))

        a_res = _gacc_get(w_res) if (want_w and cin != 3) else None
        if side_bwd is not None:
            side_bwd.wait_stream(main)
            with torch.cuda.stream(side_bwd):
                s_gw = s_gx = None
                if want_w:
                    s_gw = hb.conv2d_bwd_weight(xs, gz3, tuple(wrp.shape), 1, 0, prec, out_scale=wsc, accumulate_into=a_res)
                if want_x:
                    s_gx = res_dgrad()
                side_out = (s_gw, s_gx)
            gz3.record_stream(side_bwd)
            xs.record_stream(side_bwd)
        if downsample:
            gb3 = gsum3
            n = ctx.y2_shape[1]
            if s2d:
                _, wb2 = hb.pack_weight_s2d(w3, scale=c if alg else None)
                gxb = hb.conv2d_bwd_data(gz3, None, tuple(xb.shape), 1, 1, prec, packed=wb2, w_shape=(w3.shape[0], 4 * n, 3, 3),
                                         s2d_c=n)
                if want_w:
                    a3 = _gacc_get(w3)
                    gw3 = hb.conv2d_bwd_weight_s2d(xb, gz3, tuple(w3.shape), prec, out_scale=wsc, accumulate_into=a3)
                    if a3 is None and need[7]:
                        _gacc_put(w3, gw3)
                gz2 = hb.blur3x3_s2d_bwd(gxb, gate=y2, gate_mask=m2)  # blur adjoint + LeakyReLU derivative of y2, one pass
            else:
                wb3 = hb.pack_weight(w3, False, True, prec, scale=c)[1] if alg else None
                gxb = hb.conv2d_bwd_data(gz3, w3, tuple(xb.shape), 2, 1, prec, packed=wb3, w_shape=tuple(w3.shape))
                if want_w:
                    a3 = _gacc_get(w3)
                    gw3 = hb.conv2d_bwd_weight(xb, gz3, tuple(w3.shape), 2, 1, prec, out_scale=wsc, accumulate_into=a3)
                    if a3 is None and need[7]:
                        _gacc_put(w3, gw3)
                gz2 = hb.blur3x3_bwd_gate(gxb, y2)
        else:
            gz2 = hb.bias_act_bwd(gz3, y2)
        # bias gradients = per-channel sums of gz2 / gz1: taken from the weight-gradient kernel (which stages those
        # tensors anyway) where it can, else a read-only reduction pass
        fuse_b = want_b and want_w and os.environ.get("STYLEX_WGRAD_BIAS", "1") != "0"
        acc_b2 = False
        if want_w:
            a2w, a2b = _gacc_get(w2), _gacc_get(w2, "b")
            gw2 = hb.conv2d_bwd_weight(y1, gz2, tuple(w2.shape), 1, 1, prec, want_bias_sum=fuse_b, accumulate_into=a2w,
                                       accumulate_bias_into=a2b if a2w is not None else None)
            if fuse_b:
                gw2, gb2 = gw2
                if gb2 is True:  # added into the first use's bias gradient
                    gb2, acc_b2 = None, True
            if a2w is None:  # (only what this node really hands to the engine may be named)
                if need[5]:
                    _gacc_put(w2, gw2)
                if need[6]:
                    _gacc_put(w2, gb2, "b")
        if want_b and gb2 is None and not acc_b2:
            gb2 = _channel_sum(gz2)
        gz1 = hb.conv2d_bwd_data(gz2, w2, tuple(y1.shape), 1, 1, prec, gate=y1, gate_mask=m1)  # + LeakyReLU derivative of y1
        if getattr(ctx, "keep_gz", False):  # hand-driven backward (gp_tangent): the pre-activation gradients are reused
            ctx.gz = (gz3, gz2, gz1, alg)
        gxs = None
        acc_b1 = False
        if want_w:
            a1w = _gacc_get(w1) if cin != 3 else None  # (the padded-RGB block returns a slice of its gradient)
            a1b = _gacc_get(w1, "b") if a1w is not None else None
            gw1 = hb.conv2d_bwd_weight(x, gz1, tuple(w1p.shape), 1, 1, prec, want_bias_sum=fuse_b, accumulate_into=a1w,
                                       accumulate_bias_into=a1b)
            if fuse_b:
                gw1, gb1 = gw1
                if gb1 is True:
                    gb1, acc_b1 = None, True
            if a1w is None and cin != 3:
                if need[3]:
                    _gacc_put(w1, gw1)
                if need[4]:
                    _gacc_put(w1, gb1, "b")
        if want_b and gb1 is None and not acc_b1:
            gb1 = _channel_sum(gz1)
        if want_x:
            pk1 = dict(packed=hb.pack_weight(w1p, False, True, prec, owner=w1)[1], w_shape=tuple(w1p.shape)) \
                if (cin == 3 and isinstance(w1, torch.nn.Parameter)) else {}
            gx = hb.conv2d_bwd_data(gz1, w1p, tuple(x.shape), 1, 1, prec, **pk1)
        if side_bwd is not None:  # join: the residual-path gradients were issued on the companion stream above
            gw_res, gxs = side_out
            main.wait_stream(side_bwd)
            for t in (gw_res, gxs):
                if t is not None:
                    t.record_stream(main)
        else:
            if want_w:
                gw_res = hb.conv2d_bwd_weight(xs, gz3, tuple(wrp.shape), 1, 0, prec, out_scale=wsc, accumulate_into=a_res)
            if want_x:
                gxs = res_dgrad()
        # (round 5: the 1/sqrt(2) owed by gw_res, gw3 and gsum3 rides their reduce launches — it was a multi-tensor multiply
        # of 26 us per block, 0.47 ms per step)
        if want_w and a_res is None and cin != 3 and need[1]:
            _gacc_put(w_res, gw_res)
        if want_w and cin == 3:
            gw1, gw_res = gw1[:, :3].contiguous(), gw_res[:, :3].contiguous()
        if want_x:
            if downsample:
                hb.add_at_even_(gx, gxs)  # adjoint of the even-pixel gather, summed in place
            else:
                gx += gxs
            if cin == 3:
                gx = gx[:, :3]
        return (gx, gw_res if need[1] else None, gb_res if need[2] else None, gw1 if need[3] else None,
                gb1 if need[4] else None, gw2 if need[5] else None, gb2 if need[6] else None,
                gw3 if need[7] else None, gb3 if need[8] else None, None)


# Gradient accumulation inside the weight-gradient launches (round 5).  A parameter used by several fast-path nodes of ONE
# backward pass (the encoder of a generator phase: E(x) and E(G(x)), reference stylex_train.py:1383-1395) used to hand the
# autograd engine one gradient tensor per use, which the engine sums with an add launch per parameter (117 per step).  Now
# the first node to run hands the engine its tensor and notes it through a WEAK reference (hb.RawGrad: a strong one would keep
# AccumulateGrad from stealing the tensor; the weak one also tells if the engine dropped it for an out-of-place sum); a later node of the same graph task adds into that tensor in its reduce launch
# (stylex_conv2d_bwd_weight_ex, accumulate) and returns None.  The engine keeps the first tensor alive, unmodified, until
# the parameter's AccumulateGrad node runs — after every use, by its dependency count.  Same two rounded fp32 operations as
# the engine's add: bit-identical gradients (tests/test_hip_parity.py::test_twice_used_block_accumulates_in_the_reduce_launch).
#
# Stream ordering (round 6, ADVICE medium).  The engine knows nothing of the later node's write, so the two orderings it would
# have provided are made explicit: (1) a node running on ANOTHER HIP stream than the tensor's producer (E(x) on the caller's
# stream, E(G(x)) on a Trainer._fork side stream) first makes its stream wait for the producer's; (2) the stream it added on
# is joined into the stream backward() was called from by an engine callback at the end of the pass (what the engine's own
# leaf-stream sync does for gradients it knows of), and GradSync's in-backward bucket launches wait for it as well
# (parallel.py::_launch reads _GACC_STREAMS).  tests/test_hip_parity.py::test_twice_used_block_on_two_streams.
_GACC = {}
_GACC_TASK = [-2]
_GACC_ON = os.environ.get("STYLEX_GRAD_ACC", "1") != "0"
_GACC_STREAMS = set()  # streams that added into a gradient produced on another stream, during the current backward pass
_GACC_WAITED = {}  # (waiting stream, producer stream) -> production number up to which the wait already holds
_GACC_SEQ = [0]


def _gacc_join():
    """End of the backward pass (engine callback, on the thread and stream that called backward())."""
    if _GACC_STREAMS:
        cur = torch.cuda.current_stream()
        for st in _GACC_STREAMS:
            if st != cur:
                cur.wait_stream(st)
        _GACC_STREAMS.clear()


def _gacc_get(w, tag=""):
    """The gradient tensor an earlier node of this backward pass produced for parameter w (tag: "b" = its layer's bias)."""
    if not _GACC_ON or not w.is_cuda:
        return None
    task = torch._C._current_graph_task_id()
    if task < 0:
        return None
    if task != _GACC_TASK[0]:
        _GACC.clear()
        _GACC_STREAMS.clear()
        _GACC_WAITED.clear()
        _GACC_TASK[0] = task
    slot = _GACC.get((w.data_ptr(), tag))
    if slot is not None and not slot.alive():  # the engine no longer holds that tensor (replaced by an out-of-place sum)
        del _GACC[(w.data_ptr(), tag)]
        return None
    if slot is not None:
        cur = torch.cuda.current_stream()
        if slot.stream != cur:
            # the tensor's producer ran on another stream: its kernel must have finished.  One wait covers every tensor that
            # stream had produced by then (slots are numbered in production order), so a whole encoder costs one event
            key = (cur, slot.stream)
            if _GACC_WAITED.get(key, 0) <= slot.seq:
                cur.wait_stream(slot.stream)
                _GACC_WAITED[key] = _GACC_SEQ[0]
            if not _GACC_STREAMS:
                torch.autograd.Variable._execution_engine.queue_callback(_gacc_join)
            _GACC_STREAMS.add(cur)
    return slot


def _gacc_put(w, g, tag=""):
    if _GACC_ON and g is not None and w.is_cuda and torch._C._current_graph_task_id() == _GACC_TASK[0] and g.is_contiguous():
        slot = _GACC[(w.data_ptr(), tag)] = hb.RawGrad(g)
        slot.seq = _GACC_SEQ[0]
        _GACC_SEQ[0] += 1


def _channel_sum(t, scale=1.0):
    """per-channel sum over (b, h, w) of scale * t in fp32 (bias gradient): read-only reduction kernel, fixed order."""
    if _reducible(t.shape[1]):
        return hb.act_bwd_reduce(t, None, False, scale, want_dx=False, want_sum=True)[1]
    r = t.sum(dim=(0, 2, 3), dtype=torch.float32)
    return r if scale == 1.0 else r * scale


# ------------------------------------------------------------------------------------------
# the HIP implementation object
# ------------------------------------------------------------------------------------------


def _pad_rgb(x, weight):
    """RGB tensors have 3 channels; the vector (16-byte) load paths of the kernels want multiples of 4.
    Zero-pad C_in 3->4 (input + weight) and N 3->4 (weight; the caller slices the output): exact."""
    n_out = weight.shape[0]
    if weight.shape[1] == 3:
        extra = 5 if x.dtype == torch.bfloat16 else 1  # 16-byte slots: 8 bf16 or 4 fp32 channels
        x = torch.cat([x, x.new_zeros(x.shape[0], extra, x.shape[2], x.shape[3])], dim=1)
        weight = torch.cat([weight, weight.new_zeros(weight.shape[0], extra, weight.shape[2], weight.shape[3])], dim=1)
    if n_out == 3:
        weight = torch.cat([weight, weight.new_zeros(1, *weight.shape[1:])], dim=0)
    return x, weight, n_out


class HipOps:
    """Functional surface used by the network modules (networks.py)."""

    name = "hip"

    @staticmethod
    def conv2d(x, weight, bias=None, stride=1, padding=0, lrelu=False, residual=None, res_scale=1.0):
        """nn.Conv2d (+ LeakyReLU(0.2)) — reference :724-736, :771, :881; with `residual` the block
        merge (conv + bias + residual) * res_scale of :743 is fused into the same kernel."""
        x, weight, n_out = _pad_rgb(_act(x), weight)
        residual = _act(residual)
        padded_out = n_out != weight.shape[0]
        if stride == 2 and padding == 0 and tuple(weight.shape[2:]) == (1, 1) and x.shape[1] % 4 == 0:
            # 1x1 / stride 2 (conv_res): contiguous 1x1 / stride-1 GEMMs over the gathered even pixels
            x, stride = _Subsample2.apply(x), 1
        if not padded_out:
            fn = _ConvBiasActFast if fast_enabled() else _ConvBiasActDD
            return fn.apply(x, weight, bias, residual, stride, padding, lrelu, res_scale)
        y = _Conv.apply(x, weight, stride, padding)
        if padded_out:
            y = y[:, :n_out]
        if lrelu:
            assert residual is None
            return _BiasAct.apply(y, bias)
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        if residual is not None:
            y = (y + residual) * res_scale
        return y

    @staticmethod
    def modulated_conv2d(x, style, weight, demod=True, eps=1e-8, coeffs=None):
        """Conv2DMod.forward (:647-667) without materialising per-sample weights:
        y = d[b,o] * conv(x * (style+1)[b,i], W),  d = rsqrt(((style+1)^2) @ sum_k W^2 + eps).
        `coeffs` = mod_coeffs(style, weight, demod, eps) when the caller computed them ahead (side stream)."""
        x = _act(x)
        s1, d_pre = coeffs if coeffs is not None else mod_coeffs(style, weight, demod, eps)
        k = weight.shape[2]
        pad = (k - 1) // 2  # _get_same_padding for stride 1, dilation 1 (:644-645)
        n_out = weight.shape[0]
        if fast_enabled() and n_out == 3 and k == 1 and not demod and hb.torgb_ok(x):
            return _ToRGBFast.apply(x, s1, weight)[:, :3]  # to-RGB: a 3-column GEMM is an HBM stream, not MFMA work
        w_run = weight
        if n_out == 3:
            w_run = torch.cat([weight, weight.new_zeros(1, *weight.shape[1:])], dim=0)
        if fast_enabled() and w_run.shape[0] % 4 == 0:
            d = d_pre
            if demod and n_out == 3:
                d = torch.cat([d, d.new_ones(d.shape[0], 1)], dim=1)
            y = _ModConvFast.apply(x, s1, d, w_run, None, None, None, pad, False)
            return y[:, :3] if n_out == 3 else y
        y = _Conv.apply(x * s1[:, :, None, None], w_run, 1, pad)
        if n_out == 3:
            y = y[:, :3]
        if demod:
            y = y * d_pre[:, :, None, None]
        return y

    @staticmethod
    def modconv_noise_act(x, style, weight, inoise, noise_w, noise_b, demod=True, eps=1e-8, coeffs=None):
        """lrelu(Conv2DMod(x, style) + noise) of GeneratorBlock (:696-714) as one fused kernel when no
        double backward can be requested, else the differentiable composition."""
        x = _act(x)
        if fast_enabled() and weight.shape[0] % 4 == 0:
            s1, d = coeffs if coeffs is not None else mod_coeffs(style, weight, demod, eps)
            plane = inoise[:, :, :, 0]
            nat = _natural_noise(inoise) if (x.is_cuda and os.environ.get("STYLEX_NOISE_NAT", "1") != "0") else None
            return _ModConvFast.apply(x, s1, d, weight, plane, noise_w, noise_b, (weight.shape[2] - 1) // 2, True, nat)
        return HipOps.noise_act(HipOps.modulated_conv2d(x, style, weight, demod, eps, coeffs=coeffs), inoise, noise_w,
                                noise_b)

    @staticmethod
    def noise_act(x, inoise, noise_w, noise_b):
        """lrelu(x + noise) with noise[b,c,h,w] = inoise[b,w,h,0]*noise_w[c] + noise_b[c]
        (the (0,3,2,1) permute of :696-698 is a spatial transpose)."""
        h, w = x.shape[2], x.shape[3]
        plane = inoise[:, :h, :w, 0].transpose(1, 2)  # [B, h(w-index), ...] -> value at (h,w) = inoise[b,w,h]
        n = plane[:, None, :, :] * noise_w.view(1, -1, 1, 1) + noise_b.view(1, -1, 1, 1)
        return _BiasAct.apply(_act(x + n), None)

    @staticmethod
    def upsample2x(x):
        return _Up.apply(_act(x))

    @staticmethod
    def blur3x3(x):
        return _Blur.apply(_act(x))

    @staticmethod
    def blur_down(x, weight, bias, residual, res_scale):
        """(conv3x3_s2(blur(x)) + bias + residual) * res_scale — the tail of DiscriminatorBlock (:733-743)."""
        b, c, h, w = x.shape
        if (_PRECISION != hb.F32 and c % 64 == 0 and weight.shape[0] % 4 == 0 and h % 2 == 0
                and w % 2 == 0 and w // 2 >= 16 and h // 2 >= 16 and tuple(weight.shape[2:]) == (3, 3)):
            x2 = _BlurS2D.apply(_act(x))
            if fast_enabled():
                return _DownS2DFast.apply(x2, weight, bias, _act(residual), res_scale)
            return _ConvBiasActDD.apply(x2, weight, bias, _act(residual), 1, 1, False, res_scale, c)
        return HipOps.conv2d(HipOps.blur3x3(x), weight, bias, stride=2, padding=1, residual=residual,
                             res_scale=res_scale)

    @staticmethod
    def rgb_block(x, prev_rgb, style, weight, upsample):
        """Whole RGBBlock (reference :618-629) on the streaming kernels: to-RGB (1x1 modulated conv, no demodulation),
        skip add, bilinear x2 and blur — two launches per direction.  The chain keeps the 4-channel storage of the
        to-RGB kernel (channel 3 is zero); returns None when the composable path must run."""
        if not fast_enabled() or os.environ.get("STYLEX_RGB_TAIL", "1") == "0":
            return None
        x = _act(x)
        if tuple(weight.shape[2:]) != (1, 1) or weight.shape[0] != 3 or not hb.torgb_ok(x):
            return None
        if prev_rgb is not None and (prev_rgb.shape[1] != 4 or prev_rgb.dtype != x.dtype):
            return None
        rgb = _ToRGBFast.apply(x, style + 1, weight)
        if upsample:
            return _RGBTailFast.apply(rgb, prev_rgb)
        return rgb if prev_rgb is None else rgb + prev_rgb

    @staticmethod
    def residual_merge(x, res):
        return (x + res) * (1 / math.sqrt(2))

    @staticmethod
    def dblock(x, conv_res, conv1, conv2, down):
        """Whole DiscriminatorBlock as one fused autograd node, or None when the composable path must run (a double
        backward may be requested, or STYLEX_DBLOCK=0)."""
        if not fast_enabled() or os.environ.get("STYLEX_DBLOCK", "1") == "0":
            return None
        if conv1.weight.shape[0] % 4 != 0:
            return None
        return _DBlockFast.apply(x, conv_res.weight, conv_res.bias, conv1.weight, conv1.bias, conv2.weight, conv2.bias,
                                 down.weight if down is not None else None, down.bias if down is not None else None,
                                 down is not None)

    @staticmethod
    def rowwise_sumsq(x2d):
        return _RowSumSq.apply(x2d)


_IMPL = HipOps


def use_impl(impl):
    """Install another implementation object (tests only: a CPU test double)."""
    global _IMPL
    prev = _IMPL
    _IMPL = impl
    return prev


def impl():
    return _IMPL


class _EqualLinearFast(torch.autograd.Function):
    """lrelu(x @ (W * lr_mul)^T + b * lr_mul) — one EqualLinear (+ LeakyReLU) of the mapping network (reference
    :576-601) as one autograd node: the scaled parameters come from a per-version cache, forward = one addmm + one
    in-place LeakyReLU, backward = activation derivative, two GEMMs, one scale, one column sum (the ATen composition
    was 4-5 launches forward and 8-9 backward per layer).  First-order only: the mapping network is never
    differentiated twice (the path-length penalty differentiates G with respect to its styles, reference :306-316)."""

    @staticmethod
    def forward(ctx, x, w, b, lr_mul, slope):
        ws, bs = hb.scaled_linear_params(w, b, lr_mul)
        y = torch.addmm(bs, x, ws.t()) if bs is not None else torch.mm(x, ws.t())
        if slope is not None:
            y = F.leaky_relu_(y, slope)
        ctx.save_for_backward(x, ws, y if slope is not None else None)
        ctx.cfg = (lr_mul, slope, b is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, ws, y = ctx.saved_tensors
        lr_mul, slope, has_b = ctx.cfg
        if y is not None:
            g = torch.ops.aten.leaky_relu_backward(g, y, slope, True)
        gx = torch.mm(g, ws) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            gs = g * lr_mul  # d(W * lr_mul) / dW: scale the (small) activation gradient once instead of both results
            gw = torch.mm(gs.t(), x) if ctx.needs_input_grad[1] else None
            gb = gs.sum(0) if has_b and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None


def equal_linear(x, weight, bias, lr_mul, slope=None):
    """EqualLinear (+ LeakyReLU(slope)) on the fused node when first-order gradients suffice, else the composition."""
    if (_IMPL is HipOps and x.is_cuda and x.dim() == 2 and fast_enabled() and x.dtype == torch.float32
            and os.environ.get("STYLEX_EQL", "1") != "0"):
        return _EqualLinearFast.apply(x, weight, bias, lr_mul, slope)
    y = F.linear(x, weight * lr_mul, bias=None if bias is None else bias * lr_mul)
    return y if slope is None else F.leaky_relu(y, slope)


class _StyleAffines(torch.autograd.Function):
    """The three affine style maps of a GeneratorBlock on ONE input — to_style1, to_style2 and to_rgb.to_style (reference
    :682-688, :609: three nn.Linear(latent, C) applied to the block's style vector) — as one GEMM each way over the
    concatenated weights (round 6).  Separately they are 3 addmm forward and, backward, 3 data-gradient GEMMs whose results
    the engine sums with two add_ launches, 3 weight-gradient GEMMs and 3 bias sums: 14 launches per block and pass, all on
    the generator's serial chain; here 1 forward (the concatenated weight is cached per parameter stamp) and 4 backward.
    The parameters stay the reference's three modules (state dict, AttFind's in-place bias edits)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, cache):
        stamp = tuple(hb._gen(t) for t in (w1, b1, w2, b2, w3, b3))
        hit = cache.get("cat")
        if hit is None or hit[0] != stamp or hit[1].device != x.device:
            hit = (stamp, torch.cat((w1.detach(), w2.detach(), w3.detach()), dim=0), torch.cat((b1.detach(), b2.detach(), b3.detach())))
            cache["cat"] = hit
        wcat, bcat = hit[1], hit[2]
        y = torch.addmm(bcat, x, wcat.t())
        ctx.save_for_backward(x, wcat)
        n1, n2, n3 = ctx.sizes = (w1.shape[0], w2.shape[0], w3.shape[0])
        # four views of ONE result: style1, style2, the to-RGB style, and [style1 | style2] = the block's style coordinates.
        # (Slicing the result outside the node cost a zero-filled full-width tensor + an add_ per slice in the backward.)
        return y[:, :n1], y[:, n1:n1 + n2], y[:, n1 + n2:], y[:, :n1 + n2]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g1, g2, g3, gc):
        x, wcat = ctx.saved_tensors
        n1, n2, n3 = ctx.sizes
        need = ctx.needs_input_grad
        z = lambda n: x.new_zeros(x.shape[0], n)  # noqa: E731  (a style nobody used)
        gy = torch.cat((g1 if g1 is not None else z(n1), g2 if g2 is not None else z(n2), g3 if g3 is not None else z(n3)), dim=1)
        if gc is not None:  # a consumer of the style coordinates (get_style_coords=True under autograd)
            gy[:, :n1 + n2] += gc
        gx = gy @ wcat if need[0] else None
        gws = gbs = (None, None, None)
        if any(need[1:7]):
            gws = (gy.t() @ x).split((n1, n2, n3), dim=0)  # row blocks of one matrix: contiguous tensors
            gbs = gy.sum(dim=0).split((n1, n2, n3))
        return (gx, gws[0] if need[1] else None, gbs[0] if need[2] else None, gws[1] if need[3] else None,
                gbs[1] if need[4] else None, gws[2] if need[5] else None, gbs[2] if need[6] else None, None)


def style_affines(istyle, lin1, lin2, lin3, cache):
    """(to_style1(w), to_style2(w), to_rgb.to_style(w), [style1 | style2]) as four views of one GEMM result (the fused node
    above), or None when the composable path must run (CPU double, double backward, non-fp32 / non-2D input)."""
    if not (_IMPL is HipOps and istyle.is_cuda and istyle.dim() == 2 and fast_enabled() and istyle.dtype == torch.float32
            and os.environ.get("STYLEX_STYLE_FUSE", "1") != "0"):
        return None
    for lin in (lin1, lin2, lin3):
        if lin.bias is None or lin.weight.dtype != torch.float32:
            return None
    return _StyleAffines.apply(istyle, lin1.weight, lin1.bias, lin2.weight, lin2.bias, lin3.weight, lin3.bias, cache)


class _ModCoeffs(torch.autograd.Function):
    """(style + 1, demodulation coefficient) of a modulated conv as ONE launch (csrc/style_coeffs.hip; `wsq`, the
    weight-only factor, is cached per parameter version) with a two-launch first-order backward — the ATen composition
    below is ~8 launches forward and ~15 backward per call, 28 calls per step."""

    @staticmethod
    def forward(ctx, style, weight, eps):
        wsq = hb.weight_sumsq(weight)
        s1, d = hb.modcoeff_fwd(style, wsq, eps)
        ctx.save_for_backward(s1, d, wsq, weight)
        return s1, d

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gs1, gd):
        s1, d, wsq, weight = ctx.saved_tensors
        need_s, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if gd is None:  # the coefficient was not used: only the direct gradient of s1 remains
            return (gs1 if need_s else None), None, None
        gstyle, gw = hb.modcoeff_bwd(gd, d, s1, wsq, weight, gs1, need_s, need_w)
        return gstyle, gw, None


def mod_coeffs(style, weight, demod=True, eps=1e-8):
    """(s+1, demodulation coefficient) of a modulated conv (:650-656 in the batched form): small dense math on
    [B,C] / [O,I] tensors, independent of the activations — the Generator evaluates it for all layers ahead of
    the conv chain on a companion stream."""
    if (demod and _IMPL is HipOps and style.is_cuda and fast_enabled() and style.dim() == 2
            and os.environ.get("STYLEX_MODCOEFF", "1") != "0"):
        return _ModCoeffs.apply(style, weight, eps)
    s1 = style + 1
    d = None
    if demod:
        wsq = weight.pow(2).sum(dim=(2, 3))  # [O, I]
        d = torch.rsqrt((s1 * s1) @ wsq.t() + eps)  # [B, O]
    return s1, d


def conv2d(*a, **k):
    return _IMPL.conv2d(*a, **k)


def modulated_conv2d(*a, **k):
    return _IMPL.modulated_conv2d(*a, **k)


def noise_act(*a, **k):
    return _IMPL.noise_act(*a, **k)


def modconv_noise_act(*a, **k):
    return _IMPL.modconv_noise_act(*a, **k)


def upsample2x(x):
    return _IMPL.upsample2x(x)


def blur3x3(x):
    return _IMPL.blur3x3(x)


def blur_down(*a, **k):
    return _IMPL.blur_down(*a, **k)


def rgb_block(*a, **k):
    fn = getattr(_IMPL, "rgb_block", None)
    return fn(*a, **k) if fn is not None else None


def residual_merge(x, res):
    return _IMPL.residual_merge(x, res)


def rowwise_sumsq(x2d):
    return _IMPL.rowwise_sumsq(x2d)
