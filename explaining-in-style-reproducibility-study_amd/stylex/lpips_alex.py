"""LPIPS (AlexNet) perceptual distance used by ``reconstruction_loss``
(reference: module-level ``lpips.LPIPS(net='alex')``, stylex/stylex_train.py:404,415;
package lpips==0.1.4, environment.yml:212).

The published metric: scale the inputs, take the five post-ReLU AlexNet feature
maps, unit-normalise each along channels, square the difference, weight with a
learned non-negative 1x1 conv per tap, average spatially and sum.  Runs on stock
PyTorch-ROCm (MIOpen) like the frozen classifier — it is a caller-side dependency
of the hot path, not one of the hand-written kernels.

Weights: ``LPIPS.load_lpips_state_dict`` accepts the key layout of the lpips
package (``net.slice*.N.weight`` / ``lin*.model.1.weight``).  Offline there is no
way to fetch them, so by default seeded random weights of the published shapes
are drawn (recorded as "parity unpinned" in DESIGN.md).
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

# (cin, cout, kernel, stride, pad, maxpool_before, torchvision-features index)
_ALEX = [(3, 64, 11, 4, 2, False, 0), (64, 192, 5, 1, 2, True, 3), (192, 384, 3, 1, 1, True, 6),
         (384, 256, 3, 1, 1, False, 8), (256, 256, 3, 1, 1, False, 10)]


class _MaxPool3s2CL(torch.autograd.Function):
    """MaxPool2d(3, 2) of a bf16 channels_last tap on csrc/frozen_ew.hip's kernels (round 6): ATen's channels_last bf16 kernel
    took 213 us per call at B = 32 (0.85 ms per train step); same maxima, same choice among equal values as ATen."""

    @staticmethod
    def forward(ctx, x):
        import hip_backend as hb

        y, idx = hb.maxpool3s2_cl_fwd(x)
        ctx.save_for_backward(idx)
        ctx.in_hw = (x.shape[2], x.shape[3])
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        import hip_backend as hb

        (idx,) = ctx.saved_tensors
        return hb.maxpool3s2_cl_bwd(gy.contiguous(memory_format=torch.channels_last), idx, ctx.in_hw)


class LPIPS(nn.Module):
    def __init__(self, net="alex", seed=4242, **_):
        super().__init__()
        if net != "alex":
            raise ValueError("only net='alex' is used by the reference")
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer("scale", torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))
        self.cw, self.cb, self.lin = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
        for (ci, co, k, s, p, mp, _) in _ALEX:
            std = (2.0 / (ci * k * k)) ** 0.5
            self.cw.append(nn.Parameter(torch.randn(co, ci, k, k, generator=g) * std, requires_grad=False))
            self.cb.append(nn.Parameter(torch.zeros(co), requires_grad=False))
            self.lin.append(nn.Parameter(torch.rand(1, co, 1, 1, generator=g) / co, requires_grad=False))
        self.eval()

    def load_lpips_state_dict(self, sd):
        with torch.no_grad():
            for i, (_, _, _, _, _, _, idx) in enumerate(_ALEX):
                self.cw[i].copy_(sd["net.slice%d.%d.weight" % (i + 1, idx)])
                self.cb[i].copy_(sd["net.slice%d.%d.bias" % (i + 1, idx)])
                self.lin[i].copy_(sd["lin%d.model.1.weight" % i])

    def _taps(self, x):
        out = []
        for i, (_, _, _, s, p, mp, _) in enumerate(_ALEX):
            if mp:
                x = F.max_pool2d(x, 3, 2)
            if i == 0 and x.is_cuda:  # the stem: its input gradient on the image-gradient kernel (frozen_resnet.first_conv)
                from frozen_resnet import first_conv

                x = F.relu(first_conv(x, self.cw[i], self.cb[i], s, p))
            else:
                x = F.relu(F.conv2d(x, self.cw[i], self.cb[i], stride=s, padding=p))
            out.append(x)
        return out

    def _taps_bf16(self, x):
        """The five taps in the bf16 speed mode (round 6): the stem on the generic implicit-GEMM kernel over the image padded to
        8 channels (its input gradient: csrc/frozen_ew.hip's image-gradient kernel; STYLEX_LPIPS_STEM=0: the library's fp32
        forward + one bridge kernel), the other four layers (5x5, then three 3x3, each + bias + ReLU in the conv epilogue) run on this library's bf16 MFMA kernels,
        first-order backward included (ops.conv2d fast path).  Returns bf16 channels_last feature maps."""
        import hip_backend as hb
        import ops
        from frozen_resnet import _ReluToCLBf16, _StemBf16, first_conv

        if os.environ.get("STYLEX_LPIPS_STEM", "1") != "0":  # the stem on the generic bf16 kernel (see _StemBf16: why)
            with hb.timing_pause():
                t = _StemBf16.apply(x, self.cw[0], self.cb[0], _ALEX[0][3], _ALEX[0][4])
        else:  # library forward; relu + cast + layout in one pass
            t = _ReluToCLBf16.apply(first_conv(x, self.cw[0], self.cb[0], _ALEX[0][3], _ALEX[0][4]))
        out = [t]
        prev = ops.set_fast(True)  # first-order gradients only ever flow through the frozen loss network
        try:
            with hb.timing_pause():  # (not StylEx convs: out of the timing hook's classes, bench.py `frozen_nets`)
                for i in range(1, len(_ALEX)):
                    _, _, _, s, p, mp, _ = _ALEX[i]
                    if mp:
                        t = _MaxPool3s2CL.apply(t) if os.environ.get("STYLEX_LPIPS_POOL", "1") != "0" else F.max_pool2d(t, 3, 2)
                    t = ops.conv2d(t, self.cw[i], self.cb[i], stride=s, padding=p, lrelu="relu")
                    out.append(t)
        finally:
            ops.set_fast(prev)
        return out

    @staticmethod
    def _bf16_path(x):
        """bf16 speed mode on the GPU with the HIP implementation installed (STYLEX_LPIPS_BF16=0: the fp32 library path)."""
        if not x.is_cuda or os.environ.get("STYLEX_LPIPS_BF16", "1") == "0":
            return False
        import ops

        return ops.get_precision() == "bf16" and ops.impl() is ops.HipOps

    def forward(self, in0, in1):
        if self._bf16_path(in0):
            f0 = self._taps_bf16((in0 - self.shift) / self.scale)
            f1 = self._taps_bf16((in1 - self.shift) / self.scale)
            lins = [w.reshape(-1) for w in self.lin]
            return _LpipsDistanceNHWC.apply(len(f0), *lins, *f0, *f1).view(-1, 1, 1, 1)
        f0 = self._taps((in0 - self.shift) / self.scale)
        f1 = self._taps((in1 - self.shift) / self.scale)
        if in0.is_cuda and f0[0].dtype == torch.float32 and os.environ.get("STYLEX_LPIPS_FUSE", "1") != "0":
            # normalise / difference / 1x1 lin / spatial mean of all five taps: one kernel per tap (csrc/frozen_ew.hip)
            lins = [w.reshape(-1) for w in self.lin]
            return _LpipsDistance.apply(len(f0), *lins, *f0, *f1).view(-1, 1, 1, 1)
        total = 0
        for i in range(len(_ALEX)):
            n0 = f0[i] / (f0[i].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            n1 = f1[i] / (f1[i].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            total = total + F.conv2d((n0 - n1) ** 2, self.lin[i]).mean(dim=(2, 3), keepdim=True)
        return total


class _LpipsDistance(torch.autograd.Function):
    """sum over taps of spatial_average(lin(normalize(f0) - normalize(f1))^2) (lpips.py forward) on the fused
    kernels; first-order gradients to the feature maps that require them (the lin weights are frozen)."""

    @staticmethod
    def forward(ctx, n, *args):
        import hip_backend as hb

        lins, f0s, f1s = args[:n], [t.contiguous() for t in args[n:2 * n]], [t.contiguous() for t in args[2 * n:]]
        need = any(t.requires_grad for t in args[n:])
        out, norms = hb.lpips_taps_fwd(f0s, f1s, lins, keep_norms=need)
        ctx.n = n
        if need:
            ctx.save_for_backward(*lins, *f0s, *f1s, *[r for pair in norms for r in pair])
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        import hip_backend as hb

        n, sv = ctx.n, ctx.saved_tensors
        lins, f0s, f1s, rs = sv[:n], sv[n:2 * n], sv[2 * n:3 * n], sv[3 * n:]
        gout = gout.contiguous().float()
        g0s, g1s = [], []
        for i in range(n):
            w0, w1 = ctx.needs_input_grad[1 + n + i], ctx.needs_input_grad[1 + 2 * n + i]
            g0, g1 = (None, None) if not (w0 or w1) else hb.lpips_tap_bwd(f0s[i], f1s[i], lins[i], rs[2 * i], rs[2 * i + 1],
                                                                          gout, w0, w1)
            g0s.append(g0)
            g1s.append(g1)
        return (None,) + (None,) * n + tuple(g0s) + tuple(g1s)


class _LpipsDistanceNHWC(torch.autograd.Function):
    """_LpipsDistance on bf16 channels_last feature maps (stylex_lpips_tap_nhwc_fwd / _bwd): the channel reductions are
    contiguous 16-byte loads instead of strided sweeps; gradients come back as bf16 channels_last, the layout the conv
    kernels' backward takes."""

    @staticmethod
    def forward(ctx, n, *args):
        import hip_backend as hb

        lins, f0s, f1s = args[:n], list(args[n:2 * n]), list(args[2 * n:])
        need = any(t.requires_grad for t in args[n:])
        out, norms = hb.lpips_taps_nhwc_fwd(f0s, f1s, lins, keep_norms=need)
        ctx.n = n
        if need:
            ctx.save_for_backward(*lins, *f0s, *f1s, *[r for pair in norms for r in pair])
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        import hip_backend as hb

        n, sv = ctx.n, ctx.saved_tensors
        lins, f0s, f1s, rs = sv[:n], sv[n:2 * n], sv[2 * n:3 * n], sv[3 * n:]
        gout = gout.contiguous().float()
        g0s, g1s = [], []
        for i in range(n):
            w0, w1 = ctx.needs_input_grad[1 + n + i], ctx.needs_input_grad[1 + 2 * n + i]
            g0, g1 = (None, None) if not (w0 or w1) else hb.lpips_tap_nhwc_bwd(f0s[i], f1s[i], lins[i], rs[2 * i], rs[2 * i + 1],
                                                                               gout, w0, w1)
            g0s.append(g0)
            g1s.append(g1)
        return (None,) + (None,) * n + tuple(g0s) + tuple(g1s)
