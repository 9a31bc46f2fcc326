"""StyleGAN2 / StylEx network modules with the reference's module API and
state-dict layout (reference: stylex/stylex_train.py:576-909), computing
through the HIP ops in ``ops.py``.

Drop-in contract: same class names, constructor arguments,
attribute names (AttFind mutates ``blocks[i].to_style{1,2}.bias`` in place),
``state_dict`` keys, and the same parameter-creation order so that
``torch.manual_seed(s)`` yields bit-identical initial weights.

Differences by design: activations are kept NHWC (channels_last) end to end;
``Conv2DMod`` never materialises per-sample weights (ops.modulated_conv2d);
bias+LeakyReLU, blur and bilinear x2 are single fused kernels.
"""
import atexit
import math
from math import log2

import os

import torch
import torch.nn.functional as F
from torch import nn

import ops


def exists(v):
    return v is not None


_SIDE = {}
atexit.register(_SIDE.clear)  # streams are released before the interpreter (and the HIP runtime) shut down


def _side_stream(t, which=0):
    """The `which`-th companion HIP stream of the current stream, for branch-level concurrency inside a network (None on
    CPU or with STYLEX_STREAMS=0).  Keyed by the current stream so that networks the Trainer itself runs
    concurrently on different streams do not meet on one shared side stream."""
    import os

    if not t.is_cuda or os.environ.get("STYLEX_STREAMS", "1") == "0":
        return None
    if torch.cuda.is_current_stream_capturing():
        # Under HIP-graph capture only the Trainer-level branch forks are used.  A block-level fork that is open at
        # the same time as a Trainer-level one (the gradient-penalty step: D(real) with its per-block companion
        # stream on the capturing stream while D(fake) runs on a Trainer side stream), or one nested inside a Trainer
        # branch, makes hipStreamEndCapture crash (ROCm 7.2; tools/graph_stage_probe.py) — the block runs inline.
        return None
    key = (t.device, torch.cuda.current_stream().cuda_stream, which)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=t.device)
    return _SIDE[key]


def leaky_relu(p=0.2):
    return nn.LeakyReLU(p, inplace=True)


class Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.shape[0], -1)


class Blur(nn.Module):
    """3x3 binomial blur, reflect border (reference :144-153).  Buffer ``f`` is kept for
    checkpoint compatibility; the taps are baked into the kernel."""

    def __init__(self):
        super().__init__()
        self.register_buffer("f", torch.Tensor([1, 2, 1]))

    def forward(self, x):
        return ops.blur3x3(x)


class Upsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) (reference :614,679)."""

    def forward(self, x):
        return ops.upsample2x(x)


class HipConv2d(nn.Conv2d):
    """nn.Conv2d parameter container (same init, same keys) running on the implicit-GEMM kernel.
    ``act=True`` fuses the LeakyReLU(0.2) that follows it in the reference's nn.Sequential."""

    def __init__(self, cin, cout, k, padding=0, stride=1, act=False):
        super().__init__(cin, cout, k, padding=padding, stride=stride)
        self.act = act

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, stride=self.stride[0], padding=self.padding[0], lrelu=self.act)


class _FusedAct(nn.Module):
    """Placeholder keeping the reference's nn.Sequential indices (the activation is fused
    into the preceding HipConv2d)."""

    def forward(self, x):
        return x


class EqualLinear(nn.Module):  # reference :576-587
    def __init__(self, in_dim, out_dim, lr_mul=1, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_dim))
        self.lr_mul = lr_mul

    def forward(self, input):
        return F.linear(input, self.weight * self.lr_mul, bias=self.bias * self.lr_mul)


class StyleVectorizer(nn.Module):  # reference :590-601
    def __init__(self, emb, depth, lr_mul=0.1):
        super().__init__()
        layers = []
        for _ in range(depth):
            layers += [EqualLinear(emb, emb, lr_mul), leaky_relu()]
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        x = F.normalize(x, dim=1)
        mods, i = list(self.net), 0
        while i < len(mods):  # EqualLinear + LeakyReLU pairs run as one fused node (ops.equal_linear)
            m = mods[i]
            if isinstance(m, EqualLinear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                x = ops.equal_linear(x, m.weight, getattr(m, "bias", None), m.lr_mul, mods[i + 1].negative_slope)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class Conv2DMod(nn.Module):  # reference :632-667
    def __init__(self, in_chan, out_chan, kernel, demod=True, stride=1, dilation=1, eps=1e-8, **kwargs):
        super().__init__()
        assert stride == 1 and dilation == 1, "only the configuration the reference uses is implemented"
        self.filters, self.demod, self.kernel = out_chan, demod, kernel
        self.stride, self.dilation, self.eps = stride, dilation, eps
        self.weight = nn.Parameter(torch.randn((out_chan, in_chan, kernel, kernel)))
        nn.init.kaiming_normal_(self.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")

    def forward(self, x, y):
        return ops.modulated_conv2d(x, y, self.weight, demod=self.demod, eps=self.eps)


class RGBBlock(nn.Module):  # reference :604-629
    def __init__(self, latent_dim, input_channel, upsample, rgba=False):
        super().__init__()
        self.input_channel = input_channel
        self.to_style = nn.Linear(latent_dim, input_channel)
        self.conv = Conv2DMod(input_channel, 4 if rgba else 3, 1, demod=False)
        self.upsample = nn.Sequential(Upsample2x(), Blur()) if upsample else None

    def forward(self, x, prev_rgb, istyle, style=None, padded=False):
        """`padded`: the caller (Generator.forward) chains the blocks on the 4-channel storage of the fused RGB path
        (channel 3 zero) and slices once at the end; otherwise the reference's 3-channel tensor is returned."""
        style = self.to_style(istyle) if style is None else style
        if padded and not self.conv.demod:
            fused = ops.rgb_block(x, prev_rgb, style, self.conv.weight, exists(self.upsample))
            if fused is not None:
                return fused
        if exists(prev_rgb) and prev_rgb.shape[1] == 4 and self.conv.weight.shape[0] == 3:
            prev_rgb = prev_rgb[:, :3]  # an earlier block took the fused path
        x = self.conv(x, style)
        if exists(prev_rgb):
            x = x + prev_rgb
        if exists(self.upsample):
            x = self.upsample(x)
        return x


class GeneratorBlock(nn.Module):  # reference :670-718
    def __init__(self, latent_dim, input_channels, filters, upsample=True, upsample_rgb=True, rgba=False):
        super().__init__()
        self.input_channels, self.filters = input_channels, filters
        self.num_style_coords = input_channels + filters
        self.upsample = Upsample2x() if upsample else None
        self.to_style1 = nn.Linear(latent_dim, input_channels)
        self.to_noise1 = nn.Linear(1, filters)
        self.conv1 = Conv2DMod(input_channels, filters, 3)
        self.to_style2 = nn.Linear(latent_dim, filters)
        self.to_noise2 = nn.Linear(1, filters)
        self.conv2 = Conv2DMod(filters, filters, 3)
        self.activation = leaky_relu()
        self.to_rgb = RGBBlock(latent_dim, filters, upsample_rgb, rgba)

    def forward_main(self, x, istyle, inoise, styles=None):
        """The feature path of the block (everything except the toRGB branch).  `styles` = (style1, style2)
        overrides the two affine style maps — the batched AttFind sweep (attfind.py) feeds per-sample offsets this
        way instead of mutating `to_style*.bias` in place as the reference notebook does."""
        if exists(self.upsample):
            x = self.upsample(x)
        coords = None
        self._rgb_style = None
        if styles is None:
            # the block's three affine style maps as one GEMM (ops._StyleAffines); the first two column blocks ARE the
            # block's style coordinates (reference :716: cat(style1, style2)), the third is handed to to_rgb
            fused = ops.style_affines(istyle, self.to_style1, self.to_style2, self.to_rgb.to_style, self.__dict__.setdefault("_aff_cache", {}))
            if fused is not None:
                style1, style2, self._rgb_style, coords = fused
            else:
                style1, style2 = self.to_style1(istyle), self.to_style2(istyle)
        else:
            style1, style2 = styles
        x = ops.modconv_noise_act(x, style1, self.conv1.weight, inoise, self.to_noise1.weight[:, 0],
                                  self.to_noise1.bias, demod=self.conv1.demod, eps=self.conv1.eps)
        x = ops.modconv_noise_act(x, style2, self.conv2.weight, inoise, self.to_noise2.weight[:, 0],
                                  self.to_noise2.bias, demod=self.conv2.demod, eps=self.conv2.eps)
        return x, (coords if coords is not None else torch.cat([style1, style2], dim=-1))

    def forward(self, x, prev_rgb, istyle, inoise):
        x, coords = self.forward_main(x, istyle, inoise)
        rgb = self.to_rgb(x, prev_rgb, istyle, style=self.__dict__.pop("_rgb_style", None))
        return x, rgb, coords


class DiscriminatorBlock(nn.Module):  # reference :721-744
    def __init__(self, input_channels, filters, downsample=True):
        super().__init__()
        self.conv_res = HipConv2d(input_channels, filters, 1, stride=(2 if downsample else 1))
        self.net = nn.Sequential(HipConv2d(input_channels, filters, 3, padding=1, act=True), _FusedAct(),
                                 HipConv2d(filters, filters, 3, padding=1, act=True), _FusedAct())
        self.downsample = nn.Sequential(Blur(), HipConv2d(filters, filters, 3, padding=1, stride=2)) if downsample else None

    def forward(self, x):
        fused = getattr(ops.impl(), "dblock", None)
        if fused is not None:  # first-order passes: the whole block is one autograd node (ops._DBlockFast)
            out = fused(x, self.conv_res, self.net[0], self.net[2], self.downsample[1] if exists(self.downsample) else None)
            if out is not None:
                return out
        # the 1x1 residual conv is independent of the two 3x3 convs until the merge: side stream
        side = _side_stream(x)
        if side is None:
            res = self.conv_res(x)
            x = self.net(x)
        else:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            x.record_stream(side)
            with torch.cuda.stream(side):
                res = self.conv_res(x)
            x = self.net(x)
            main.wait_stream(side)
            res.record_stream(main)
        if exists(self.downsample):
            down = self.downsample[1]
            # (conv_s2(blur(x)) + bias + res) / sqrt(2): blur -> space-to-depth -> halo conv with the merge of
            # :743 in its epilogue
            return ops.blur_down(x, down.weight, down.bias, res, 1 / math.sqrt(2))
        return ops.residual_merge(x, res)


def generator_filters(image_size, network_capacity, fmap_max):
    n = int(log2(image_size) - 1)
    f = [min(fmap_max, network_capacity * 2 ** (i + 1)) for i in range(n)][::-1]
    return [f[0]] + f


def discriminator_filters(image_size, network_capacity, fmap_max, transparent=False):
    n = int(log2(image_size) - 1)
    return [4 if transparent else 3] + [min(fmap_max, 4 * network_capacity * 2 ** i) for i in range(n + 1)]


class Generator(nn.Module):  # reference :747-825
    def __init__(self, image_size, latent_dim, network_capacity=16, transparent=False, attn_layers=[], no_const=False,
                 fmap_max=512):
        super().__init__()
        assert not attn_layers and not no_const, "attention / no_const variants are out of scope (SURVEY §2a)"
        self.image_size, self.latent_dim = image_size, latent_dim
        self.num_layers = int(log2(image_size) - 1)
        filters = generator_filters(image_size, network_capacity, fmap_max)
        self.no_const = no_const
        self.initial_block = nn.Parameter(torch.randn((1, filters[0], 4, 4)))
        self.initial_conv = HipConv2d(filters[0], filters[0], 3, padding=1)
        self.blocks = nn.ModuleList([])
        self.attns = nn.ModuleList([])
        for ind in range(self.num_layers):
            self.attns.append(None)
            self.blocks.append(GeneratorBlock(latent_dim, filters[ind], filters[ind + 1], upsample=ind != 0,
                                              upsample_rgb=ind != self.num_layers - 1, rgba=transparent))

    def forward(self, styles, input_noise, get_style_coords=False):
        batch = styles.shape[0]
        x = self.initial_conv(self.initial_block.expand(batch, -1, -1, -1))
        rgb, coords = None, []
        # The toRGB chain runs on the caller's stream.  (Rounds 1-5 kept an opt-in variant that ran it one block behind on a
        # side HIP stream — +0.8 % — which produced run-to-run differences: deleted in round 6.  Their very likely cause was
        # found later that round: the to-RGB data-gradient kernel itself, whenever other waves shared the GPU)
        # One UnbindBackward (a stack) instead of num_layers SelectBackwards (a zero-filled [B, L, D] tensor each, summed
        # pairwise by the engine): ~20 fewer launches per generator backward.
        per_layer = styles.unbind(1)
        for li, block in enumerate(self.blocks):
            x, sc = block.forward_main(x, per_layer[li], input_noise)
            coords.append(sc)
            rgb = block.to_rgb(x, rgb, per_layer[li], style=block.__dict__.pop("_rgb_style", None), padded=True)
        if rgb.shape[1] == 4 and self.blocks[0].to_rgb.conv.filters == 3:
            rgb = rgb[:, :3]  # fused RGB path: 4-channel storage, channel 3 is zero
        rgb = rgb.float()  # activations may be stored in bf16; the module API returns fp32 images
        if get_style_coords:
            return rgb, torch.cat(coords, dim=1)
        return rgb


class DiscriminatorE(nn.Module):  # reference :842-909 (D: 1 logit; encoder: encoder_dim outputs)
    def __init__(self, image_size, network_capacity=16, fq_layers=[], fq_dict_size=256, attn_layers=[],
                 transparent=False, encoder=False, encoder_dim=512, fmap_max=512, conditional=False):
        super().__init__()
        assert not fq_layers and not attn_layers, "fq / attention variants are out of scope (SURVEY §2a)"
        filters = discriminator_filters(image_size, network_capacity, fmap_max, transparent)
        n = len(filters) - 1
        self.blocks = nn.ModuleList([DiscriminatorBlock(filters[i], filters[i + 1], downsample=i != n - 1)
                                     for i in range(n)])
        self.attn_blocks = nn.ModuleList([None] * n)
        self.quantize_blocks = nn.ModuleList([None] * n)
        chan_last = filters[-1]
        self.final_conv = HipConv2d(chan_last, chan_last, 3, padding=1)
        self.flatten = Flatten()
        self.encoder_dim = encoder_dim
        self.encoder = encoder
        # conditional = the "new architecture" discriminator (reference stylex_train_new.py:887-916): two logits,
        # combined with the classifier probabilities of the conditioning batch (projection-style conditional GAN)
        self.conditional = conditional and not encoder
        self.fc = nn.Linear(2 * 2 * chan_last, encoder_dim if encoder else (2 if self.conditional else 1))

    def forward(self, x, probabilities=None):
        for block in self.blocks:
            x = block(x)
        x = self.final_conv(x)
        x = self.fc(self.flatten(x).float())
        if self.conditional:
            if probabilities is None:  # the reference's default argument torch.Tensor([0.0, 0.0]) (:892)
                probabilities = x.new_zeros(1, 2)
            x = x[:, 0] * probabilities[:, 0] + x[:, 1] * probabilities[:, 1]
        return x.squeeze()
