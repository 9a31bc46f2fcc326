"""Frozen torchvision-style ResNet (BasicBlock), the classifier of the StylEx step, in eval mode with frozen parameters
(reference stylex/resnet_classifier.py:29-71): its BatchNorms are affine maps.

Forward: the library's fp32 convolutions, untouched (north_star), with everything between them on fused fp32 kernels
(FusedTailResNet).  bf16 speed mode, a pass whose input gradient is wanted: the data gradient on this library's bf16 conv kernels
gated by the signs of the fp32 activations (_ResNetBodyHybrid), BatchNorm scale folded into the packed weights (_fold).
Also here: the stems' image-gradient hook (first_conv) and the bf16 stem / bridge nodes LPIPS-AlexNet uses.
"""
import torch
import torch.nn.functional as F
from torch import nn

import hip_backend as hb
import ops


def _fold(conv, bn):
    """(W', b') with bn(conv(x, W)) == conv(x, W') + b' for an eval-mode BatchNorm."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = (conv.weight * scale.view(-1, 1, 1, 1)).detach().contiguous()
    b = (bn.bias - bn.running_mean * scale).detach().contiguous()
    if conv.bias is not None:
        b = b + conv.bias.detach() * scale
    return nn.Parameter(w, requires_grad=False), nn.Parameter(b, requires_grad=False)


def _basic_block_resnet(model):
    """torchvision-style ResNet of BasicBlocks (3x3 convs, no grouped / bottleneck blocks)?"""
    try:
        blocks = [b for layer in (model.layer1, model.layer2, model.layer3, model.layer4) for b in layer]
        return all(hasattr(b, "conv1") and hasattr(b, "bn2") and not hasattr(b, "conv3") and
                   b.conv1.groups == 1 and b.conv1.kernel_size == (3, 3) for b in blocks)
    except AttributeError:
        return False


# ---------------------------------------------------------------------------------------------------------------------
# Default on the GPU: library convolutions (fp32, untouched) + fused elementwise tails (csrc/frozen_ew.hip)
# ---------------------------------------------------------------------------------------------------------------------


def _affine(bn):
    """(scale, shift) with bn(x) == x * scale[c] + shift[c] for an eval-mode BatchNorm (fp32, like ATen's own
    batch_norm_elementwise: invstd = 1 / sqrt(var + eps))."""
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float().contiguous()
    shift = (bn.bias - bn.running_mean * scale).detach().float().contiguous()
    return scale, shift


class _FirstConv(torch.autograd.Function):
    """The stem of a frozen network over the image (ResNet conv1 7x7/2, LPIPS-AlexNet 11x11/4): forward = the library's
    convolution, untouched; backward = csrc/frozen_ew.hip conv_image_grad_kernel instead of the library's dense transposed
    convolution (round 6: 0.47 / 0.52 ms -> 0.25 / 0.38 ms per call at B = 32; same fp32 arithmetic, fixed order)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad):
        ctx.cfg = (stride, pad, tuple(x.shape[2:]))
        ctx.save_for_backward(w)
        return F.conv2d(x, w, bias, stride, pad)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        (w,) = ctx.saved_tensors
        stride, pad, hw = ctx.cfg
        return hb.conv_image_grad(gy, w, hw, stride, pad), None, None, None, None


def first_conv(x, w, bias, stride, pad):
    """conv2d of a frozen stem; the input gradient on the image-gradient kernel where it applies (GPU, fp32, <= 4 image
    channels, stride 1 / 2 / 4, frozen weights).  STYLEX_IMAGE_GRAD=0: plain F.conv2d."""
    import os

    # (register-tiled kernel, alone on the GPU at B = 32: 0.25 ms against the library's 0.47 at stride 4, 0.38 against 0.52 at stride 2:
    # profiles/r06_s_probe_first_conv.txt; the first three versions ran at 1.2, 0.5 and 0.33 / 0.58 ms)
    mode = os.environ.get("STYLEX_IMAGE_GRAD", "1")
    if (x.is_cuda and x.requires_grad and x.dtype == torch.float32 and x.shape[1] <= 4 and stride in (1, 2, 4)
            and w.shape[2] == w.shape[3] <= 15 and pad < w.shape[2] and not w.requires_grad
            and (bias is None or not bias.requires_grad) and mode != "0"):
        return _FirstConv.apply(x, w, bias, stride, pad)
    return F.conv2d(x, w, bias, stride, pad)


class _ReluToCLBf16(torch.autograd.Function):
    """bf16 channels_last copy of relu(x) for a dense fp32 NCHW x, one pass each way (the hand-over from a library fp32 stem to
    the bf16 kernels: lpips_alex._taps_bf16)."""

    @staticmethod
    def forward(ctx, x):
        y = hb.nchw_to_cl_bf16(x.contiguous(), relu=True)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return hb.cl_bf16_to_nchw(gy.contiguous(memory_format=torch.channels_last), gate=y)


class _StemBf16(torch.autograd.Function):
    """relu(conv2d(x, w, b, stride, pad)) of a frozen stem over the 3-channel image in the bf16 speed mode, bf16 channels_last
    out (round 6, LPIPS-AlexNet's 11x11 / 4 layer): the image is cast + padded to one 16-byte channel slot (stylex_pad_rgb8)
    and the layer runs on the generic implicit-GEMM kernel with tap-major K (121 taps x 8 channels), bias + ReLU in its epilogue;
    backward = bridge kernel (bf16 NHWC -> fp32 NCHW, gated by the output's sign) + stylex_conv_image_grad.  Why not the
    library: its immediate mode serves this layer with an NHWC implicit-GEMM kernel on some boxes (90 us per 32 images) and with a
    per-image im2col + GEMM on others (64 launches, 1.9 ms per step: profiles/r06_p_steady_state_kernels.txt)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad):
        P = hb.BF16_ACT
        xp = hb.pad_rgb8(x)
        wp = hb.pad_in_channels(w, 8 - w.shape[1])
        owner = w if isinstance(w, torch.nn.Parameter) else None
        packed = hb.pack_weight(wp, True, False, P, owner=owner)[0]
        y = hb.conv2d_fwd(xp, wp, stride, pad, P, bias=bias, lrelu="relu", packed=packed, w_shape=tuple(wp.shape))
        ctx.save_for_backward(y, w)
        ctx.cfg = (stride, pad, tuple(x.shape[2:]))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        y, w = ctx.saved_tensors
        stride, pad, hw = ctx.cfg
        g = hb.cl_bf16_to_nchw(gy.contiguous(memory_format=torch.channels_last), gate=y)
        return hb.conv_image_grad(g, w, hw, stride, pad), None, None, None, None


class _AffineAct(torch.autograd.Function):
    """act(x * scale[c] + shift[c] (+ residual)): `bn -> relu` / `bn -> (+ identity) -> relu` of a BasicBlock
    (torchvision resnet.py BasicBlock.forward) in one pass; first-order backward to x and the residual."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        y = hb.affine_act_fwd(x, scale, shift, residual, relu)
        ctx.save_for_backward(y if relu else None, scale)
        ctx.cfg = (relu, residual is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        relu, has_res = ctx.cfg
        want_res = has_res and ctx.needs_input_grad[3]
        gx, gres = hb.affine_act_bwd(gy.contiguous(), y, scale, relu, want_res)
        return (gx if ctx.needs_input_grad[0] else None), None, None, gres, None


class _AffineReluPool(torch.autograd.Function):
    """maxpool(3, 2, 1)(relu(x * scale[c] + shift[c])): the stem of the ResNet after conv1 in one pass."""

    @staticmethod
    def forward(ctx, x, scale, shift):
        y, idx = hb.affine_relu_maxpool_fwd(x, scale, shift, want_idx=x.requires_grad)
        ctx.save_for_backward(idx, scale)
        ctx.in_hw = tuple(x.shape[2:])
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        idx, scale = ctx.saved_tensors
        return hb.affine_relu_maxpool_bwd(gy.contiguous(), idx, scale, ctx.in_hw), None, None


class FusedTailResNet(nn.Module):
    """The frozen eval-mode ResNet with its convolutions exactly as before (F.conv2d on the module's own fp32 weights)
    and every BatchNorm / ReLU / residual add / max-pool between them on the fused fp32 kernels."""

    def __init__(self, model):
        super().__init__()
        assert not model.training, "the classifier must be in eval mode (BatchNorm as an affine map)"
        self.model = model
        self.bns = [(name, mod) for name, mod in model.named_modules() if isinstance(mod, nn.BatchNorm2d)]
        self._build()

    def _stamp(self):
        return tuple(t._version for _, bn in self.bns for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))

    def _build(self):
        self.aff = {name: _affine(bn) for name, bn in self.bns}
        self.stamp = self._stamp()

    @staticmethod
    def supports(model):
        if not _basic_block_resnet(model):
            return False
        mp = model.maxpool
        as2 = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)  # noqa: E731
        return (as2(mp.kernel_size), as2(mp.stride), as2(mp.padding), as2(mp.dilation), mp.ceil_mode) == (
            (3, 3), (2, 2), (1, 1), (1, 1), False) and all(p.dtype == torch.float32 for p in model.parameters())

    @staticmethod
    def _conv(x, conv):
        return F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)

    def _bf16_backward(self, x):
        """bf16 speed mode, a pass whose input gradient is wanted (the classifier on GENERATED images: the KL term's path to
        G): forward unchanged on the library's fp32 convolutions, data gradient on this library's bf16 kernels
        (_ResNetBodyHybrid).  STYLEX_FROZEN_BWD_BF16=0 switches it off."""
        import os

        if not (x.is_cuda and x.requires_grad and torch.is_grad_enabled()) or os.environ.get("STYLEX_FROZEN_BWD_BF16", "1") == "0":
            return False
        return ops.get_precision() == "bf16" and ops.impl() is ops.HipOps

    def forward(self, x):
        m = self.model
        if self._stamp() != self.stamp:  # BatchNorm tensors were modified in place (a state dict loaded later)
            self._build()
            self._hyb = None
        x = x.float().contiguous()
        if self._bf16_backward(x):
            c1 = m.conv1
            stem = first_conv(x, c1.weight, c1.bias, c1.stride[0], c1.padding[0]) if c1.groups == 1 else self._conv(x, c1)
            p0 = _AffineReluPool.apply(stem.contiguous(), *self.aff["bn1"])
            pstamp = tuple(p._version for p in m.parameters())  # (a state dict loaded into the classifier later: refold)
            if getattr(self, "_hyb", None) is None or self._hyb.stamp != pstamp:
                self._hyb = _HybridPlan(self)
                self._hyb.stamp = pstamp
            y = _ResNetBodyHybrid.apply(p0, self)
            return m.fc(torch.flatten(m.avgpool(y), 1))
        c1 = m.conv1
        if c1.groups == 1 and c1.dilation == (1, 1) and c1.stride[0] == c1.stride[1] and c1.padding[0] == c1.padding[1]:
            stem = first_conv(x, c1.weight, c1.bias, c1.stride[0], c1.padding[0])
        else:
            stem = self._conv(x, c1)
        x = _AffineReluPool.apply(stem.contiguous(), *self.aff["bn1"])
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(m, "layer%d" % li)):
                pre = "layer%d.%d." % (li, bi)
                idt = x
                if blk.downsample is not None:
                    idt = _AffineAct.apply(self._conv(x, blk.downsample[0]).contiguous(), *self.aff[pre + "downsample.1"],
                                           None, False)
                out = _AffineAct.apply(self._conv(x, blk.conv1).contiguous(), *self.aff[pre + "bn1"], None, True)
                x = _AffineAct.apply(self._conv(out, blk.conv2).contiguous(), *self.aff[pre + "bn2"], idt, True)
        return m.fc(torch.flatten(m.avgpool(x), 1))


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: fp32 library forward, bf16 data gradient with the forward's EXACT ReLU gates
# ---------------------------------------------------------------------------------------------------------------------


class _HybridPlan:
    """Per BasicBlock: the conv weights with the following eval-BatchNorm's scale folded in (the data gradient of
    `bn(conv(x))` w.r.t. x is the data gradient of conv with W * scale[n]), as frozen Parameters so that their packed bf16
    operand copies are cached like any other weight's."""

    def __init__(self, net):
        self.blocks = []
        m = net.model
        for li in range(1, 5):
            for blk in getattr(m, "layer%d" % li):
                e = {"stride": blk.conv1.stride[0], "w1": _fold(blk.conv1, blk.bn1)[0], "w2": _fold(blk.conv2, blk.bn2)[0]}
                if blk.downsample is not None:
                    e["wd"] = _fold(blk.downsample[0], blk.downsample[1])[0]
                self.blocks.append(e)


class _ResNetBodyHybrid(torch.autograd.Function):
    """layer1 .. layer4 of the frozen eval-mode ResNet as ONE node.  forward: FusedTailResNet's own kernels (the library's fp32
    convolutions + the fused fp32 tails) — bit-identical logits; it keeps the two post-ReLU activations of every block as
    bf16 channels_last copies.  backward: the data gradient on this library's bf16 conv kernels, gated by the SIGNS of those
    fp32 activations (a bf16 copy has the sign of the fp32 value), so no ReLU gate can flip — what made a bf16 FORWARD too
    noisy a classifier signal (resnet_classifier.py) does not exist here; the error is the bf16 rounding of the gradient
    operands, ~1 % in the L2 sense (test_frozen_classifier_bf16_data_gradient).  Replaces ~20 library bwd-data convolutions, their
    layout transposes and 14 tail kernels per call with ~50 launches of kernels that run 2-3x faster on these shapes."""

    @staticmethod
    def forward(ctx, x, net):
        m = net.model
        bf = hb.nchw_to_cl_bf16  # (ATen's strided copy: 45 us per activation, 0.73 ms per call; the LDS-tile kernel: ~5)
        saved, shapes = [], []
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(m, "layer%d" % li)):
                pre = "layer%d.%d." % (li, bi)
                idt = x
                if blk.downsample is not None:
                    idt = hb.affine_act_fwd(net._conv(x, blk.downsample[0]).contiguous(), *net.aff[pre + "downsample.1"], None, False)
                h = hb.affine_act_fwd(net._conv(x, blk.conv1).contiguous(), *net.aff[pre + "bn1"], None, True)
                shapes.append((tuple(x.shape), tuple(h.shape)))
                x = hb.affine_act_fwd(net._conv(h, blk.conv2).contiguous(), *net.aff[pre + "bn2"], idt, True)
                saved += [bf(h), bf(x)]
        ctx.save_for_backward(*saved)
        ctx.net, ctx.shapes = net, shapes
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        plan, P = ctx.net._hyb.blocks, hb.BF16_ACT
        saved = ctx.saved_tensors
        g = gy.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
        with hb.timing_pause():  # (not StylEx convs: out of the timing hook's classes, bench.py `frozen_nets`)
            g = _ResNetBodyHybrid._chain(ctx, plan, saved, g, P)
        return hb.cl_bf16_to_nchw(g), None

    @staticmethod
    def _chain(ctx, plan, saved, g, P):
        ga, gb = g, None  # the gradients reaching a block's output: through the next block's conv path / its identity path
        for k in range(len(plan) - 1, -1, -1):
            e, h, out = plan[k], saved[2 * k], saved[2 * k + 1]
            x_shape, h_shape = ctx.shapes[k]
            gz = hb.relu_gate_add(ga, gb, out)  # d relu(. + identity) of the sum of both, one pass
            if "wd" in e:  # 1x1 / stride-2 projection: a 1x1 / stride-1 data gradient over the even pixels, zero-inserted
                half = (x_shape[0], x_shape[1], gz.shape[2], gz.shape[3])
                g_idt = hb.subsample2_bwd(hb.conv2d_bwd_data(gz, e["wd"], half, 1, 0, P), (x_shape[2], x_shape[3]))
            else:
                g_idt = gz
            gh = hb.conv2d_bwd_data(gz, e["w2"], h_shape, 1, 1, P, gate=h, gate_slope=0.0)  # + d relu of conv1's output
            ga, gb = hb.conv2d_bwd_data(gh, e["w1"], x_shape, e["stride"], 1, P), g_idt
        return ga + gb
