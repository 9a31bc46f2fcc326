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
(``torch.channels_last``), fp32.

The module-level functions (``conv2d`` …) dispatch to an *implementation
object*.  The default and only product implementation is the HIP one below; it
raises if the library or a GPU is missing.  ``tests/`` may install a CPU test
double with ``use_impl`` to exercise the host logic without a GPU.
"""
import math

import torch

import hip_backend as hb

_PRECISION = hb.F32


def set_precision(name):
    """'fp32' -> exact f32 MFMA (parity mode); 'bf16' -> bf16 MFMA operands, fp32 accumulate."""
    global _PRECISION
    _PRECISION = {"fp32": hb.F32, "f32": hb.F32, "bf16": hb.BF16}[name]


def get_precision():
    return "bf16" if _PRECISION == hb.BF16 else "fp32"


# ------------------------------------------------------------------------------------------
# convolution triad
# ------------------------------------------------------------------------------------------


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad):
        x = hb.to_cl(x)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, _PRECISION)
        return hb.conv2d_fwd(x, w, stride, pad, _PRECISION)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad, _ = ctx.cfg
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _Dgrad.apply(gy, w, tuple(x.shape), stride, pad)
        if ctx.needs_input_grad[1]:
            gw = _Wgrad.apply(x, gy, tuple(w.shape), stride, pad)
        return gx, gw, None, None


class _Dgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, w, x_shape, stride, pad):
        gy = hb.to_cl(gy)
        ctx.save_for_backward(gy, w)
        ctx.cfg = (x_shape, stride, pad)
        return hb.conv2d_bwd_data(gy, w, x_shape, stride, pad, _PRECISION)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        x_shape, stride, pad = ctx.cfg
        d_gy = d_w = None
        if ctx.needs_input_grad[0]:
            d_gy = _Conv.apply(ggx, w, stride, pad)
        if ctx.needs_input_grad[1]:
            d_w = _Wgrad.apply(ggx, gy, tuple(w.shape), stride, pad)
        return d_gy, d_w, None, None, None


class _Wgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gy, w_shape, stride, pad):
        x, gy = hb.to_cl(x), hb.to_cl(gy)
        ctx.save_for_backward(x, gy)
        ctx.cfg = (w_shape, stride, pad)
        return hb.conv2d_bwd_weight(x, gy, w_shape, stride, pad, _PRECISION)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        w_shape, stride, pad = ctx.cfg
        d_x = d_gy = None
        ggw = ggw.contiguous()
        if ctx.needs_input_grad[0]:
            d_x = _Dgrad.apply(gy, ggw, tuple(x.shape), stride, pad)
        if ctx.needs_input_grad[1]:
            d_gy = _Conv.apply(x, ggw, stride, pad)
        return d_x, d_gy, None, None, None


# ------------------------------------------------------------------------------------------
# bias + LeakyReLU(0.2)
# ------------------------------------------------------------------------------------------


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        x = hb.to_cl(x)
        y = hb.bias_act_fwd(x, bias)
        ctx.save_for_backward(y)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gx = _BiasActBwd.apply(gy, y)
        gb = gx.sum(dim=(0, 2, 3)) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        return gx, gb


class _BiasActBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, y):
        gy = hb.to_cl(gy)
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
        return hb.upsample2x_fwd(hb.to_cl(x))

    @staticmethod
    def backward(ctx, gy):
        return _UpBwd.apply(gy)


class _UpBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy):
        return hb.upsample2x_bwd(hb.to_cl(gy))

    @staticmethod
    def backward(ctx, ggx):
        return _Up.apply(ggx)


class _Blur(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return hb.blur3x3_fwd(hb.to_cl(x))

    @staticmethod
    def backward(ctx, gy):
        return _BlurBwd.apply(gy)


class _BlurBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy):
        return hb.blur3x3_bwd(hb.to_cl(gy))

    @staticmethod
    def backward(ctx, ggx):
        return _Blur.apply(ggx)


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
# the HIP implementation object
# ------------------------------------------------------------------------------------------


class HipOps:
    """Functional surface used by the network modules (networks.py)."""

    name = "hip"

    @staticmethod
    def conv2d(x, weight, bias=None, stride=1, padding=0, lrelu=False):
        """nn.Conv2d (+ LeakyReLU(0.2)) — reference :724-736, :771, :881."""
        y = _Conv.apply(x, weight, stride, padding)
        if lrelu:
            return _BiasAct.apply(y, bias)
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        return y

    @staticmethod
    def modulated_conv2d(x, style, weight, demod=True, eps=1e-8):
        """Conv2DMod.forward (:647-667) without materialising per-sample weights:
        y = d[b,o] * conv(x * (style+1)[b,i], W),  d = rsqrt(((style+1)^2) @ sum_k W^2 + eps)."""
        s1 = style + 1
        k = weight.shape[2]
        pad = (k - 1) // 2  # _get_same_padding for stride 1, dilation 1 (:644-645)
        y = _Conv.apply(x * s1[:, :, None, None], weight, 1, pad)
        if demod:
            wsq = weight.pow(2).sum(dim=(2, 3))  # [O, I]
            d = torch.rsqrt((s1 * s1) @ wsq.t() + eps)  # [B, O]
            y = y * d[:, :, None, None]
        return y

    @staticmethod
    def noise_act(x, inoise, noise_w, noise_b):
        """lrelu(x + noise) with noise[b,c,h,w] = inoise[b,w,h,0]*noise_w[c] + noise_b[c]
        (the (0,3,2,1) permute of :696-698 is a spatial transpose)."""
        h, w = x.shape[2], x.shape[3]
        plane = inoise[:, :h, :w, 0].transpose(1, 2)  # [B, h(w-index), ...] -> value at (h,w) = inoise[b,w,h]
        n = plane[:, None, :, :] * noise_w.view(1, -1, 1, 1) + noise_b.view(1, -1, 1, 1)
        return _BiasAct.apply(x + n, None)

    @staticmethod
    def upsample2x(x):
        return _Up.apply(x)

    @staticmethod
    def blur3x3(x):
        return _Blur.apply(x)

    @staticmethod
    def residual_merge(x, res):
        return (x + res) * (1 / math.sqrt(2))

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


def conv2d(*a, **k):
    return _IMPL.conv2d(*a, **k)


def modulated_conv2d(*a, **k):
    return _IMPL.modulated_conv2d(*a, **k)


def noise_act(*a, **k):
    return _IMPL.noise_act(*a, **k)


def upsample2x(x):
    return _IMPL.upsample2x(x)


def blur3x3(x):
    return _IMPL.blur3x3(x)


def residual_merge(x, res):
    return _IMPL.residual_merge(x, res)


def rowwise_sumsq(x2d):
    return _IMPL.rowwise_sumsq(x2d)
