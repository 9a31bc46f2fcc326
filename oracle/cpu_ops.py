"""TEST INFRASTRUCTURE — CPU test double for the product's op surface.

Implements the functional interface of ``stylex/ops.py`` (``HipOps``) with the
oracle's literal pure-torch formulas so that ``tests/`` can exercise the
product's HOST logic (module wiring, init order, Trainer control flow, gradient
all-reduce) on a machine without a GPU.  Installed only by tests through
``ops.use_impl(CpuOracleOps)``; the product never imports this file.
"""
import math

import torch
import torch.nn.functional as F

import stylex_oracle as so


class CpuOracleOps:
    name = "cpu-oracle-test-double"

    @staticmethod
    def conv2d(x, weight, bias=None, stride=1, padding=0, lrelu=False, residual=None, res_scale=1.0):
        y = F.conv2d(x, weight, bias, stride=stride, padding=padding)
        if residual is not None:
            y = (y + residual) * res_scale
        return so.lrelu(y) if lrelu else y

    @staticmethod
    def modconv_noise_act(x, style, weight, inoise, noise_w, noise_b, demod=True, eps=1e-8, coeffs=None):
        return CpuOracleOps.noise_act(so.modulated_conv2d(x, style, weight, demod, eps), inoise, noise_w, noise_b)

    @staticmethod
    def modulated_conv2d(x, style, weight, demod=True, eps=1e-8, coeffs=None):
        return so.modulated_conv2d(x, style, weight, demod, eps)

    @staticmethod
    def noise_act(x, inoise, noise_w, noise_b):
        h, w = x.shape[2], x.shape[3]
        crop = inoise[:, :h, :w, :]
        n = (crop * noise_w.view(1, 1, 1, -1) + noise_b.view(1, 1, 1, -1)).permute(0, 3, 2, 1)
        return so.lrelu(x + n)

    @staticmethod
    def upsample2x(x):
        return so.upsample2x_bilinear(x)

    @staticmethod
    def blur3x3(x):
        return so.blur3x3_reflect(x)

    @staticmethod
    def blur_down(x, weight, bias, residual, res_scale):
        return CpuOracleOps.conv2d(so.blur3x3_reflect(x), weight, bias, stride=2, padding=1, residual=residual,
                                   res_scale=res_scale)

    @staticmethod
    def residual_merge(x, res):
        return (x + res) * (1 / math.sqrt(2))

    @staticmethod
    def rowwise_sumsq(x2d):
        return x2d.pow(2).sum(dim=1)
