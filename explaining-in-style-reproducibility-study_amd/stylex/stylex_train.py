"""StylEx model container, losses and Trainer — drop-in for the reference's
``stylex/stylex_train.py`` (default architecture, cli.py:17), running the
adversarial step on the HIP kernels behind ``ops.py``.

Kept verbatim from the reference's API: ``Trainer`` kwargs,
methods and attributes; ``StylEx`` children/optimisers; the free functions the
AttFind notebook imports.  New, MI355X-first behaviour:
  * no import-time CUDA assert, no LPIPS pinned to cuda:0 (reference :51,:404);
  * D-phase generator/encoder forwards run under no_grad (their graph is never
    used, :1330-1331) and D weight-gradients are skipped in the G phase (they
    are zeroed by the next D_opt.zero_grad, :1297) — results identical;
  * the three encoder-step backward calls (:1436-1438) are one backward of the sum;
  * loss scalars stay on the device; one host sync per phase instead of 5-8
    ``.item()`` per micro-step;
  * data-parallel training = one process per GPU with RCCL all-reduce of flat
    gradient buckets (``parallel.py``), including the encoder the reference forgot
    to wrap (:1188-1193).
"""
import atexit
import threading
import json
import math
import multiprocessing
import os
from math import floor, log2
from pathlib import Path
from random import random
from shutil import rmtree

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.optim import Adam
from torch.utils import data
from torch.utils.data.distributed import DistributedSampler  # noqa: F401  (re-exported for the notebook)

import ops
import parallel
from lpips_alex import LPIPS
from mobilenet_classifier import MobileNet
from networks import (Blur, Conv2DMod, DiscriminatorBlock, DiscriminatorE, EqualLinear, Flatten,  # noqa: F401
                      Generator, GeneratorBlock, HipConv2d, RGBBlock, StyleVectorizer, Upsample2x, exists, leaky_relu)
from resnet_classifier import ResNet
from version import __version__

NUM_CORES = multiprocessing.cpu_count()
EXTS = ["jpg", "jpeg", "png"]


class NanException(Exception):
    pass


# ------------------------------------------------------------------------------------------
# small helpers with the reference's names
# ------------------------------------------------------------------------------------------


def default(value, d):
    return value if exists(value) else d


def cycle(iterable):
    while True:
        for i in iterable:
            yield i


def cast_list(el):
    return el if isinstance(el, list) else [el]


def is_empty(t):
    if isinstance(t, torch.Tensor):
        return t.nelement() == 0
    return not exists(t)


def set_requires_grad(model, flag):
    for p in model.parameters():
        p.requires_grad = flag


def _cat(ts):
    return ts[0] if len(ts) == 1 else torch.cat(ts, dim=0)


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _own_rng_mode(fn):
    """Public Trainer entry points draw with THEIR Trainer's device_rng setting, whatever another Trainer of the
    process used last (the module-level draw helpers read _Staging.DEVICE_RNG)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if fn.__name__ != "train":
            self._drain_draw_ahead()  # a prefetched draw of the next train() call must not interleave with these draws
        prev, _Staging.DEVICE_RNG = _Staging.DEVICE_RNG, self.device_rng
        try:
            return fn(self, *a, **k)
        finally:
            _Staging.DEVICE_RNG = prev
    return wrapped


def raise_if_nan(t):  # reference :269-271
    if torch.isnan(t):
        raise NanException


def gradient_accumulate_contexts(gradient_accumulate_every, is_ddp, ddps):
    """Micro-step iterator of the reference (:274-285): under DDP every micro-step but the last runs without
    gradient exchange.  `ddps` are objects with a ``no_sync()`` context manager — torch DDP wrappers, or this package's
    ``parallel.GradSync`` (whose collectives only ever run after the last micro-step: its no_sync is a null context)."""
    from contextlib import ExitStack, nullcontext

    for i in range(gradient_accumulate_every):
        last = i == gradient_accumulate_every - 1
        with ExitStack() as stack:
            if is_ddp and not last:
                for d in ddps:
                    stack.enter_context(d.no_sync() if hasattr(d, "no_sync") else nullcontext())
            yield


def loss_backwards(fp16, loss, optimizer, loss_id, **kwargs):  # reference :288-293 (apex fp16 is out of scope)
    assert not fp16, "apex fp16 is out of scope; use ops.set_precision('bf16')"
    loss.backward(**kwargs)


def _dev(device):
    """Accept the reference's integer rank, a torch.device or None."""
    if isinstance(device, torch.device):
        return device
    if device is None or not torch.cuda.is_available():
        return torch.device("cpu")
    return torch.device("cuda:%d" % int(device))


class _Staging:
    """Reusable pinned host buffers for the per-step RNG uploads (latents, image noise).  A pageable
    .to(device) blocks the host until the stream drains (the copy is stream-ordered), serialising host and GPU
    at the top of every phase; drawing straight into a pinned buffer and copying with non_blocking=True does
    not.  Each (shape) has a small ring of buffers; a slot is reused only after the event recorded behind its
    last copy has completed."""
    RING = 8
    _slots = {}
    _upload_streams = {}

    @classmethod
    def take(cls, shape):
        ring = cls._slots.setdefault(tuple(shape), {"i": 0, "bufs": []})
        if len(ring["bufs"]) < cls.RING:
            ring["bufs"].append([torch.empty(shape, pin_memory=True), None])
            slot = ring["bufs"][-1]
        else:
            slot = ring["bufs"][ring["i"] % cls.RING]
            ring["i"] += 1
            if slot[1] is not None:
                slot[1].synchronize()
        return slot

    ENV_DEVICE_RNG = os.environ.get("STYLEX_DEVICE_RNG", "0") == "1"
    DEVICE_RNG = ENV_DEVICE_RNG  # set by the Trainer that is drawing (per-instance state, see _own_rng_mode)
    _tls = threading.local()     # .device_rng: per-thread override (the draw-ahead worker always draws on the CPU)

    @classmethod
    def upload(cls, shape, fill, device):
        """fill(buf) draws in place on the CPU generator (same stream consumption as torch.randn/empty+fill).
        DEVICE_RNG (STYLEX_DEVICE_RNG=1 / Trainer(device_rng=True)): draw on the GPU generator instead — no 9 ms host
        draw and no 8 MB upload per noise plane, but a different random stream than the reference's CPU one (the
        parity default keeps the CPU draw order)."""
        device = _dev(device)
        if device.type == "cuda" and getattr(cls._tls, "device_rng", cls.DEVICE_RNG):
            return fill(torch.empty(shape, device=device))
        if device.type != "cuda":
            return fill(torch.empty(shape)).to(device)
        slot = cls.take(shape)
        if _capturing() or os.environ.get("STYLEX_UPLOAD_STREAM", "1") == "0":
            out = fill(slot[0]).to(device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
            return out
        # The copy goes to a dedicated upload stream: the host runs whole phases ahead of the GPU, so a copy enqueued
        # on the compute stream sits behind that backlog and then executes IN ORDER between two kernels (0.35-0.4 ms
        # per 8 MB noise plane, four per step, nothing else running); on its own stream the SDMA engine moves it while
        # the compute kernels ahead of it are still running, and the consumer only waits for the (long finished) event.
        cur = torch.cuda.current_stream(device)
        up = cls._upload_streams.get(device)
        if up is None:
            up = cls._upload_streams[device] = torch.cuda.Stream(device)
        host = fill(slot[0])
        with torch.cuda.stream(up):
            out = host.to(device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(up)
        cur.wait_event(slot[1])
        out.record_stream(cur)
        return out


def _release_staging_at_exit():
    """Pinned buffers and their events are freed before interpreter teardown (see hip_backend._release_at_exit)."""
    try:
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except Exception:  # noqa: BLE001
        pass
    _Staging._slots.clear()
    _Staging._upload_streams.clear()


atexit.register(_release_staging_at_exit)


def noise(n, latent_dim, device):  # reference :319-320 — CPU RNG, then copy (keeps the draw order)
    return _Staging.upload((n, latent_dim), lambda t: t.normal_(), device)


def noise_list(n, layers, latent_dim, device):
    return [(noise(n, latent_dim, device), layers)]


def mixed_list(n, layers, latent_dim, device):
    tt = int(torch.rand(()).numpy() * layers)
    return noise_list(n, tt, latent_dim, device) + noise_list(n, layers - tt, latent_dim, device)


def latent_to_w(style_vectorizer, latent_descr, probabilities=None):
    """reference :332-333; with `probabilities` the new architecture's form (stylex_train_new.py:332-333): the
    classifier probabilities of the conditioning batch are appended to every mapped latent."""
    zs = [z for z, _ in latent_descr]
    if len(zs) > 1 and zs[0].is_cuda and all(z.shape[1:] == zs[0].shape[1:] for z in zs):
        # style mixing maps 2 latents: one pass of the (row-independent) mapping network over their concatenation
        ws = style_vectorizer(torch.cat(zs, dim=0)).split([z.shape[0] for z in zs], dim=0)
    else:
        ws = [style_vectorizer(z) for z in zs]
    if probabilities is None:
        return [(w, num_layers) for w, (_, num_layers) in zip(ws, latent_descr)]
    return [(torch.cat((w, probabilities), dim=1), num_layers) for w, (_, num_layers) in zip(ws, latent_descr)]


def image_noise(n, im_size, device):  # reference :336-337
    return _Staging.upload((n, im_size, im_size, 1), lambda t: t.uniform_(0., 1.), device)


def evaluate_in_chunks(max_batch_size, model, *args):
    chunks = list(zip(*[a.split(max_batch_size, dim=0) for a in args]))
    outs = [model(*c) for c in chunks]
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)


def styles_def_to_tensor(styles_def):
    return torch.cat([t[:, None, :].expand(-1, n, -1) for t, n in styles_def], dim=1)


def slerp(val, low, high):
    low_n = low / torch.norm(low, dim=1, keepdim=True)
    high_n = high / torch.norm(high, dim=1, keepdim=True)
    omega = torch.acos((low_n * high_n).sum(1))
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


def make_weights_for_balanced_classes(dataset, nclasses):  # imported by the AttFind notebook
    count = [0] * nclasses
    for _, label in dataset:
        count[int(label)] += 1
    total = float(sum(count))
    per_class = [total / c if c else 0.0 for c in count]
    return [per_class[int(label)] for _, label in dataset]


# ------------------------------------------------------------------------------------------
# losses (reference :296-316, :370-438)
# ------------------------------------------------------------------------------------------


def _flat_rows(t):
    """[B, ...] -> [B, n] without a layout copy for channels_last tensors (a sum of squares
    does not care about the element order)."""
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)
    return t.reshape(t.shape[0], -1)


def gradient_penalty(images, output, weight=10):
    """10 * mean((||dD/dx||_2 - 1)^2) on reals (reference :296-303); the per-sample norm is the
    wavefront-shuffle reduction kernel."""
    return weight * ((gradient_norms(images, output) - 1) ** 2).mean()


def gradient_norms(images, output):
    """Per-sample ||dD/dx||_2 (the inner term of gradient_penalty), differentiable."""
    with ops.inputs_only():  # parameter gradients of this first backward would be discarded
        (gradients,) = torch.autograd.grad(outputs=output, inputs=images, grad_outputs=torch.ones_like(output),
                                           create_graph=True, retain_graph=True, only_inputs=True)
    return ops.rowwise_sumsq(_flat_rows(gradients)).sqrt()


def draw_pl_noise(shape, device):
    """N(0,1)/sqrt(H*W) image noise of calc_pl_lengths (:308-309), drawn on the CPU generator (draw-order parity)."""
    num_pixels = shape[2] * shape[3]
    return _Staging.upload(tuple(shape), lambda t: t.normal_().div_(math.sqrt(num_pixels)), device)


def calc_pl_lengths(styles, images, pl_noise=None):
    """Path lengths (reference :306-316).  `pl_noise` lets the caller draw the noise itself (the Trainer draws it
    micro-step by micro-step in the reference's order and evaluates the batched micro-steps in one call: the
    per-sample lengths are independent)."""
    if pl_noise is None:
        pl_noise = draw_pl_noise(images.shape, images.device)
    outputs = (images * pl_noise).sum()
    with ops.inputs_only():
        (pl_grads,) = torch.autograd.grad(outputs=outputs, inputs=styles, grad_outputs=torch.ones_like(outputs),
                                          create_graph=True, retain_graph=True, only_inputs=True)
    return ops.pl_lengths(pl_grads)  # (pl_grads ** 2).sum(dim=2).mean(dim=1).sqrt()


# The scalar reductions below are one forward and one backward launch each on the GPU (csrc/losses.hip,
# ops._Hinge / _KLLogits / _L1Mean / _PLLengths); on other devices the reference's torch composition.
def gen_hinge_loss(fake, real):
    return ops.gen_hinge_loss(fake)  # fake.mean()


def hinge_loss(real, fake):
    return ops.hinge_loss(real, fake)  # (F.relu(1 + real) + F.relu(1 - fake)).mean()


def lpips_normalize(images):
    flat = images.reshape(images.shape[0], -1)
    n = flat.shape[1]
    if flat.is_cuda and n % 256 == 0:
        # per-sample extrema in two levels: a [B, n] row reduction runs on B workgroups only (160 us per call at
        # 256 px); [B * 256, n / 256] rows fill the GPU.  Same values; the gradient still goes to one extremal element
        rows = flat.view(flat.shape[0], 256, n // 256)
        hi = rows.max(dim=2)[0].max(dim=1)[0].view(-1, 1, 1, 1)
        lo = rows.min(dim=2)[0].min(dim=1)[0].view(-1, 1, 1, 1)
    else:
        hi = flat.max(dim=1)[0].view(-1, 1, 1, 1)
        lo = flat.min(dim=1)[0].view(-1, 1, 1, 1)
    return (images - lo) / (hi - lo) * 2 - 1


_LPIPS = {}


def get_lpips(device):
    """Lazily built per device (the reference builds it on cuda:0 at import, :404)."""
    key = str(device)
    if key not in _LPIPS:
        _LPIPS[key] = LPIPS(net="alex").to(device)
    return _LPIPS[key]


def _frozen_layout(images):
    """Memory layout handed to the frozen library networks (classifier, LPIPS): dense NCHW — MIOpen's fp32 Winograd kernels
    beat its NHWC implicit-GEMM ones on these shapes (measured: +2 % on the whole step), and the generator output arrives
    channels_last."""
    return images.contiguous()


def perceptual_loss(encoder_batch, generated_images, lpips_fn=None):
    lpips_fn = lpips_fn or get_lpips(encoder_batch.device)
    return lpips_fn(_frozen_layout(lpips_normalize(encoder_batch)),
                    _frozen_layout(lpips_normalize(generated_images))).mean()


def reconstruction_loss(encoder_batch, generated_images, generated_images_w, encoder_w, lpips_fn=None, perceptual=None):
    """Reference :426-438.  `perceptual` may be passed in when the caller evaluated the LPIPS branch itself
    (the Trainer runs it on a side stream)."""
    if perceptual is None:
        perceptual = perceptual_loss(encoder_batch, generated_images, lpips_fn)
    return 0.1 * perceptual + 0.1 * ops.l1_mean(encoder_w, generated_images_w) + 1 * ops.l1_mean(encoder_batch,
                                                                                               generated_images)


def classifier_kl_loss(real_classifier_logits, fake_classifier_logits):
    return ops.kl_logits(real_classifier_logits, fake_classifier_logits)


# ------------------------------------------------------------------------------------------
# dataset (reference :520-556) — PIL only, torchvision is optional
# ------------------------------------------------------------------------------------------


def resize_geometry(w, h, s):
    """Size after the reference's ``resize_to_minimum_size`` + ``transforms.Resize(s)`` (:480-483, :535-536) on a PIL
    image, i.e. torchvision 0.11.1 ``functional_pil.resize`` with an int size: the SHORTER side becomes s, the longer
    one int(s * long / short) (truncated, not rounded), and an image whose shorter side already is s is returned
    untouched.  (The minimum-size step resizes a too-small image the same way, after which Resize is the identity.)"""
    short, long = (w, h) if w <= h else (h, w)
    if short == s:
        return w, h
    new_long = int(s * long / short)
    return (s, new_long) if w <= h else (new_long, s)


def center_crop_offsets(w, h, s):
    """(left, top) of ``transforms.CenterCrop(s)``: torchvision computes int(round((side - s) / 2.0)) with Python's
    round-half-to-even (a 37-wide image is cropped from column 2, a 35-wide one from column 2 as well)."""
    return int(round((w - s) / 2.0)), int(round((h - s) / 2.0))


def random_resized_crop_box(w, h, scale=(0.5, 1.0), ratio=(0.98, 1.02)):
    """(top, left, height, width) of ``transforms.RandomResizedCrop.get_params`` (torchvision 0.11.1) with the
    reference's arguments (:537): up to 10 attempts, each drawing an area fraction and a log-uniform aspect ratio from
    torch's GLOBAL generator (``torch.empty(1).uniform_``), the first box that fits gets ``torch.randint`` offsets;
    otherwise the central fallback box."""
    area = h * w
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        cw = int(round(math.sqrt(target_area * aspect)))
        ch = int(round(math.sqrt(target_area / aspect)))
        if 0 < cw <= w and 0 < ch <= h:
            top = torch.randint(0, h - ch + 1, size=(1,)).item()
            left = torch.randint(0, w - cw + 1, size=(1,)).item()
            return top, left, ch, cw
    in_ratio = float(w) / float(h)
    if in_ratio < min(ratio):
        cw, ch = w, int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        ch, cw = h, int(round(h * max(ratio)))
    else:
        cw, ch = w, h
    return (h - ch) // 2, (w - cw) // 2, ch, cw


class Dataset(data.Dataset):
    """Reference ``Dataset`` (:520-547) without torchvision: the transform chain is spelled out on PIL — mode
    conversion (:443-452), Resize, ``RandomApply(aug_prob, RandomResizedCrop(scale=(0.5, 1), ratio=(0.98, 1.02)),
    CenterCrop)`` (ONE Python ``random()`` per item even at aug_prob 0, :96 — with ``num_workers=0``, the DDP
    default :1238, those draws interleave with the Trainer's own), ToTensor (bytes / 255 in float32).  After the mode
    conversion the channel count always matches, so ``expand_greyscale`` (:455-477) is the identity.  Pinned by
    tests/golden/dataset_items.npz, captured from the reference class."""

    def __init__(self, folder, image_size, transparent=False, aug_prob=0.):
        super().__init__()
        self.folder, self.image_size, self.transparent, self.aug_prob = folder, image_size, transparent, aug_prob
        self.paths = [p for ext in EXTS for p in Path(f"{folder}").glob(f"**/*.{ext}")]
        assert len(self.paths) > 0, f"No images were found in {folder} for training"

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        from PIL import Image

        img = Image.open(self.paths[index])
        mode = "RGBA" if self.transparent else "RGB"
        if img.mode != mode:
            img = img.convert(mode)
        s = self.image_size
        size = resize_geometry(*img.size, s)
        if size != img.size:
            img = img.resize(size, Image.BILINEAR)
        w, h = img.size
        if random() < self.aug_prob:
            top, left, ch, cw = random_resized_crop_box(w, h)
            img = img.crop((left, top, left + cw, top + ch)).resize((s, s), Image.BILINEAR)
        else:
            left, top = center_crop_offsets(w, h, s)
            img = img.crop((left, top, left + s, top + s))
        arr = np.array(img, dtype=np.uint8)
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class MNIST_1vA(data.Dataset):
    """Named-dataset hook of the reference (:495-517); needs torchvision's MNIST, absent offline."""

    def __init__(self, folder="./", digit=8):
        raise RuntimeError("MNIST_1vA needs torchvision.datasets.MNIST (download) — unavailable offline")


class AugWrapper(nn.Module):  # reference :558-571
    def __init__(self, D, image_size):
        super().__init__()
        self.D = D

    def forward(self, images, prob=0., types=[], detach=False, probabilities=None):
        if random() < prob:  # one Python random() per call even at prob 0 (RNG parity)
            from diff_augment import DiffAugment

            if not (0.5 > random()):
                images = torch.flip(images, dims=(3,))
            images = DiffAugment(images, types=types)
        if detach:
            images = images.detach()
        if probabilities is not None:  # conditional discriminator (stylex_train_new.py:564-572)
            return self.D(images, probabilities=probabilities)
        return self.D(images)


# ------------------------------------------------------------------------------------------
# model container (reference :912-999)
# ------------------------------------------------------------------------------------------


class StylEx(nn.Module):
    def __init__(self, image_size, latent_dim=514, fmap_max=512, style_depth=8, network_capacity=16, transparent=False,
                 fp16=False, cl_reg=False, steps=1, lr=1e-4, ttur_mult=2, fq_layers=[], fq_dict_size=256,
                 attn_layers=[], no_const=False, lr_mlp=0.1, rank=0, classifier_labels=2, encoder_class=None,
                 kl_rec_during_disc=False, capturable=False, conditional=False):
        super().__init__()
        assert not fp16 and not cl_reg and encoder_class is None, "apex fp16 / cl_reg / debug encoders: out of scope"
        self.lr, self.steps, self.ema_beta, self.fp16 = lr, steps, 0.995, False
        # conditional = the reference's second shipped architecture (stylex_train_new.py:922-969): the mapping
        # network works on latent_dim - 2 dimensions (the 2 classifier probabilities are appended to W), D has two
        # logits weighted by those probabilities, and the encoder trains at a fixed lr of 1e-5
        self.conditional, self.num_classes = conditional, 2
        s_dim = latent_dim - self.num_classes if conditional else latent_dim
        # construction ORDER is part of the contract: it fixes the RNG stream of the initial weights
        self.encoder = DiscriminatorE(image_size, network_capacity, encoder=True, fq_layers=fq_layers,
                                      fq_dict_size=fq_dict_size, attn_layers=attn_layers, transparent=transparent,
                                      fmap_max=fmap_max)
        self.S = StyleVectorizer(s_dim, style_depth, lr_mul=lr_mlp)
        self.G = Generator(image_size, latent_dim, network_capacity, transparent=transparent, attn_layers=attn_layers,
                           no_const=no_const, fmap_max=fmap_max)
        self.D = DiscriminatorE(image_size, network_capacity, fq_layers=fq_layers, fq_dict_size=fq_dict_size,
                                attn_layers=attn_layers, transparent=transparent, fmap_max=fmap_max,
                                conditional=conditional)
        self.SE = StyleVectorizer(s_dim, style_depth, lr_mul=lr_mlp)
        self.GE = Generator(image_size, latent_dim, network_capacity, transparent=transparent, attn_layers=attn_layers,
                            no_const=no_const, fmap_max=fmap_max)  # the reference omits fmap_max here (:937-938) and
        # then crashes in reset_parameter_averaging for any fmap_max that actually caps a layer; identical at 512
        self.D_cl = None
        self.D_aug = AugWrapper(self.D, image_size)
        set_requires_grad(self.SE, False)
        set_requires_grad(self.GE, False)
        generator_params = list(self.G.parameters()) + list(self.S.parameters()) + list(self.encoder.parameters())
        if conditional:  # stylex_train_new.py:967-969: encoder in its own parameter group at lr 1e-5
            generator_params = [{"params": list(self.G.parameters()) + list(self.S.parameters())},
                                {"params": list(self.encoder.parameters()), "lr": 1e-5}]
        self.G_opt = Adam(generator_params, lr=self.lr, betas=(0.5, 0.9))
        self.D_opt = Adam(self.D.parameters(), lr=self.lr * ttur_mult, betas=(0.5, 0.9))
        self._init_weights()
        self.reset_parameter_averaging()
        self.to(_dev(rank))
        opt_kw = {}
        if _dev(rank).type == "cuda" and ops.get_precision() == "bf16":
            # speed mode: one fused multi-tensor Adam launch per optimiser instead of ~23 foreach launches (same
            # update rule; the fp32 parity mode keeps the default implementation the goldens were pinned with)
            opt_kw["fused"] = True
        if _dev(rank).type == "cuda" and capturable:
            opt_kw["capturable"] = True  # step counters live on the device: the step can sit inside a HIP graph
        if opt_kw:
            self.G_opt = Adam(generator_params, lr=self.lr, betas=(0.5, 0.9), **opt_kw)
            self.D_opt = Adam(self.D.parameters(), lr=self.lr * ttur_mult, betas=(0.5, 0.9), **opt_kw)

    def _init_weights(self):
        for m in self.modules():  # the reference tests exact nn.Conv2d / nn.Linear types (:975-977)
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")
        for block in self.G.blocks:
            for lin in (block.to_noise1, block.to_noise2):
                nn.init.zeros_(lin.weight)
                nn.init.zeros_(lin.bias)

    def EMA(self):
        for ma, cur in ((self.SE, self.S), (self.GE, self.G)):
            for p_cur, p_ma in zip(cur.parameters(), ma.parameters()):
                p_ma.data = p_ma.data * self.ema_beta + (1 - self.ema_beta) * p_cur.data
                # `.data =` rebinds the storage without touching the version counter, and the allocator may hand back
                # the address of an older average: stamp the parameter so no cached operand copy of it can match
                p_ma._stylex_gen = getattr(p_ma, "_stylex_gen", 0) + 1

    def reset_parameter_averaging(self):
        self.SE.load_state_dict(self.S.state_dict())
        self.GE.load_state_dict(self.G.state_dict())

    def forward(self, x):
        return x


# ------------------------------------------------------------------------------------------
# Trainer (reference :1002-1774)
# ------------------------------------------------------------------------------------------


def _lazy_loss(name):
    """Loss scalar attribute that resolves the step's pending device->host copy on first read."""
    def get(self):
        self._resolve_losses(raise_nan=False)
        return self.__dict__.get("_" + name)

    def put(self, v):
        self.__dict__["_" + name] = v

    return property(get, put)


class Trainer:
    # Same attribute names as the reference (:1108-1116), backed by ONE asynchronous device->host copy per
    # step: train() does not wait for the GPU, so the host prepares the next step's inputs while the GPU
    # still runs this one.  Reading any of them (print_log, tests) waits for the copy and nothing else.
    d_loss = _lazy_loss("d_loss")
    g_loss = _lazy_loss("g_loss")
    total_rec_loss = _lazy_loss("total_rec_loss")
    total_kl_loss = _lazy_loss("total_kl_loss")
    last_gp_loss = _lazy_loss("last_gp_loss")
    _pending = None
    def __init__(self, name="default", results_dir="results", models_dir="models", base_dir="./", image_size=128,
                 network_capacity=16, fmap_max=512, transparent=False, batch_size=4, mixed_prob=0.9,
                 gradient_accumulate_every=1, lr=2e-4, lr_mlp=0.1, ttur_mult=2, rel_disc_loss=False, num_workers=None,
                 save_every=1000, evaluate_every=1000, num_image_tiles=8, trunc_psi=0.6, fp16=False, cl_reg=False,
                 no_pl_reg=False, fq_layers=[], fq_dict_size=256, attn_layers=[], no_const=False, aug_prob=0.,
                 aug_types=["translation", "cutout"], top_k_training=False, generator_top_k_gamma=0.99,
                 generator_top_k_frac=0.5, dual_contrast_loss=False, dataset_aug_prob=0., calculate_fid_every=None,
                 calculate_fid_num_images=12800, clear_fid_cache=False, is_ddp=False, rank=0, world_size=1, log=False,
                 kl_scaling=1, rec_scaling=10, classifier_path="mnist.pth", num_classes=2, encoder_class=None,
                 alternating_training=True, sample_from_encoder=False, dataset_name=None, tensorboard_dir=None,
                 classifier_name=None,
                 # --- extensions (defaults reproduce the reference) ---
                 classifier=None, lpips_fn=None, gp_every=4, pl_every=32, pl_after=5000, device=None,
                 save_training_state=False, graphs=None, graph_warmup=4, new_architecture=False, device_pipeline=None,
                 device_rng=None, *args, **kwargs):
        kl_rec_during_disc = kwargs.pop("kl_rec_during_disc", False)  # cli.py forwards it; only the new architecture reads it
        # new_architecture = the conditional-D variant the reference ships as stylex_train_new.py (cli.py:17-22)
        self.new_architecture = bool(new_architecture)
        # Round 5: checked against the reference itself (profiles/r05_kl_rec_during_disc_reference.txt).  With
        # kl_rec_during_disc=True the reference's own Trainer.train() raises on the first encoder micro-step of the
        # discriminator phase — "Trying to backward through the graph a second time" at stylex_train_new.py:1408: rec_loss
        # and kl_loss share the generator's graph and the first backward (:1403) frees it.  There is no behaviour to match,
        # so the option is rejected here as well, with the reference's own failure as the reason.
        if self.new_architecture and kl_rec_during_disc:
            raise RuntimeError("kl_rec_during_disc=True cannot run in the reference either: stylex_train_new.py:1403-1408 "
                               "backwards twice through one graph (RuntimeError on the first encoder micro-step of the "
                               "discriminator phase); see profiles/r05_kl_rec_during_disc_reference.txt")
        self.model_params = [args, kwargs]
        self.StylEx = None
        self.kl_scaling, self.rec_scaling = kl_scaling, rec_scaling
        self.alternating_training = alternating_training
        if self.new_architecture and alternating_training:
            # stylex_train_new.py:1166-1171: the x2 of the alternating schedule is applied once here instead of per loss
            self.rec_scaling *= 2
            self.kl_scaling *= 2
        self.name = name
        base_dir = Path(base_dir)
        self.base_dir = base_dir
        self.results_dir = base_dir / results_dir
        self.models_dir = base_dir / models_dir
        self.fid_dir = base_dir / "fid" / name
        self.config_path = self.models_dir / name / ".config.json"
        assert log2(image_size).is_integer(), "image size must be a power of 2 (64, 128, 256, 512, 1024)"
        assert not fp16, "apex fp16 is out of scope; use ops.set_precision('bf16')"
        assert not (dual_contrast_loss or top_k_training or rel_disc_loss or cl_reg), "variant losses are out of scope"
        self.image_size, self.network_capacity, self.fmap_max = image_size, network_capacity, fmap_max
        self.transparent = transparent
        self.fq_layers, self.fq_dict_size = cast_list(fq_layers), fq_dict_size
        self.attn_layers, self.no_const = cast_list(attn_layers), no_const
        self.aug_prob, self.aug_types = aug_prob, aug_types
        self.lr, self.lr_mlp, self.ttur_mult = lr, lr_mlp, ttur_mult
        self.batch_size, self.num_workers, self.mixed_prob = batch_size, num_workers, mixed_prob
        self.num_image_tiles, self.evaluate_every, self.save_every = num_image_tiles, evaluate_every, save_every
        self.steps = 0
        self.av = None
        self.trunc_psi = trunc_psi
        self.no_pl_reg = no_pl_reg
        self.pl_mean = None
        self.gradient_accumulate_every = gradient_accumulate_every
        self.fp16 = False
        self.d_loss = self.g_loss = self.total_rec_loss = self.total_kl_loss = 0
        self.q_loss = self.last_gp_loss = self.last_cr_loss = self.last_fid = None
        self.init_folders()
        self.loader = None
        self.dataset = None
        self.dataset_aug_prob = dataset_aug_prob
        self.calculate_fid_every, self.calculate_fid_num_images = calculate_fid_every, calculate_fid_num_images
        self.clear_fid_cache = clear_fid_cache
        self.is_ddp, self.is_main, self.rank, self.world_size = is_ddp, rank == 0, rank, world_size
        self.sample_from_encoder = sample_from_encoder
        self.logger = None
        self.gp_every, self.pl_every, self.pl_after = gp_every, pl_every, pl_after
        if os.environ.get("STYLEX_DETERMINISTIC") == "1":
            # The HIP path is bit-reproducible run to run (fixed-order reductions, also across its streams); the only
            # noise source of a step is MIOpen's default algorithm choice for the frozen classifier / LPIPS.  Pinning
            # it makes a whole training run reproducible for about 1 % of step throughput.
            torch.backends.cudnn.deterministic = True
        # MIOpen algorithm choice for the frozen classifier / LPIPS convolutions: the reference's setting, immediate
        # mode (cli.py:38 cudnn.benchmark = False) — also what bench.py measures.  STYLEX_MIOPEN_BENCHMARK=1 lets
        # MIOpen search instead; PyTorch then asks for an EXHAUSTIVE search, which on this ROCm 7.2 / gfx950 stack runs
        # tuning candidates of ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC that fault on small batches (see the
        # head of hip_backend.py: that solver is disabled by default), and measured no faster than immediate mode (818 vs 805-814
        # images/s, profiles/r04_c_ab_miopen_modes.txt).
        if os.environ.get("STYLEX_MIOPEN_BENCHMARK") == "1":
            torch.backends.cudnn.benchmark = True
        self.save_training_state = save_training_state
        self.device = _dev(device if device is not None else rank)
        # whole-step HIP graphs (see _graphs_enabled); the first `graph_warmup` train() calls run eagerly so that
        # both step shapes, MIOpen's algorithm search and the allocator have been exercised before the capture
        self.graphs = (os.environ.get("STYLEX_GRAPHS", "0") == "1") if graphs is None else bool(graphs)
        self.graph_warmup = graph_warmup
        self._static, self._graph_cache, self._graph_pool, self._calls, self._graph_warm = {}, {}, None, 0, set()
        # backward-time raise_if_nan hook (reference :1352): opt-in, and never under DDP — it would raise on ONE rank
        # from inside a backward whose bucket all-reduces the other ranks are already waiting in; there the MAX-reduced
        # device-side flag of train() is the (collective) NaN check
        self._nan_hook = os.environ.get("STYLEX_NAN_HOOK", "0") == "1" and not is_ddp
        # N2 (input_pipeline.py): decode-only workers + prefetch thread + on-device resize / crop / scaling
        self.device_pipeline = (os.environ.get("STYLEX_DEVICE_PIPELINE", "0") == "1") if device_pipeline is None \
            else bool(device_pipeline)
        # per-instance (a class-level switch leaked into every later Trainer of the process and broke their CPU-RNG parity)
        # Under data parallelism the latents / noise planes are drawn on the GPU generator by default: every rank shares
        # ONE host with the others (an 8-GPU node leaves each rank 1/8 of the cores), the CPU-generator draws of the four
        # noise planes cost ~13 ms of host time per step, and there is no reference random stream to keep — the
        # reference's multi-GPU path never worked (README.md:81) and the ranks draw from rank-distinct seeds anyway.
        # Single-GPU runs keep the reference's CPU draw order (the parity default).  STYLEX_DEVICE_RNG=0 / 1 or
        # Trainer(device_rng=...) override.
        if device_rng is None:
            env = os.environ.get("STYLEX_DEVICE_RNG")
            device_rng = (env == "1") if env is not None else (bool(is_ddp) and torch.cuda.is_available())
        self.device_rng = bool(device_rng)
        # Draw-ahead (round 4): the CPU-generator draws of a phase (latents, the image-noise planes: ~14 ms of host time
        # per call at 256 px) run on ONE worker thread while the main thread enqueues the previous phase's kernels —
        # the draws of a train() call never depend on its compute, and a single worker issues them in exactly the
        # reference's order (every golden step test runs with it).  1 (GPU default) = within a call: the generator
        # phase's draws under the discriminator phase; 2 (opt-in) = also the next call's discriminator-phase draws under
        # this call's generator phase (skipped when this call ends with evaluate / save / FID) — that changes what a
        # CALLER's own draws between two train() calls see (the evaluation-surface golden, which calls evaluate()
        # between steps, fails with it), so it is not a default; 0 = off.  Off with device RNG, augmentation
        # (AugWrapper draws inside the compute) and graphs.
        self._draw_mode = int(os.environ.get("STYLEX_DRAW_AHEAD", "1" if self.device.type == "cuda" else "0"))
        self._draw_worker, self._next_d = None, None
        self.lpips_fn = lpips_fn
        self.num_classes = num_classes
        # the classifier wrappers keep the reference's `cuda_rank` argument (it passes the process rank, :1157-1160, which
        # is the GPU index only on one node with one rank per GPU in order); the Trainer hands them the index of ITS device
        dev_index = self.device.index if (self.device.type == "cuda" and self.device.index is not None) else rank
        if classifier is not None:
            self.classifier = classifier
        elif str(classifier_name).lower() == "resnet":
            self.classifier = ResNet(classifier_path, cuda_rank=dev_index, output_size=num_classes, image_size=image_size)
        else:
            self.classifier = MobileNet(classifier_path, cuda_rank=dev_index, output_size=num_classes, image_size=image_size)
        self.tb_writer = None
        if exists(tensorboard_dir):
            try:
                from torch.utils.tensorboard import SummaryWriter

                self.tb_writer = SummaryWriter(os.path.join(tensorboard_dir, name))
            except Exception:  # tensorboard is optional (absent offline)
                self.tb_writer = None

    # ---- bookkeeping -------------------------------------------------------------------

    @property
    def image_extension(self):
        return "jpg" if not self.transparent else "png"

    @property
    def checkpoint_num(self):
        return floor(self.steps // self.save_every)

    @property
    def hparams(self):
        return {"image_size": self.image_size, "network_capacity": self.network_capacity}

    def init_StylEx(self):
        args, kwargs = self.model_params
        self.StylEx = StylEx(lr=self.lr, lr_mlp=self.lr_mlp, ttur_mult=self.ttur_mult, image_size=self.image_size,
                             network_capacity=self.network_capacity, fmap_max=self.fmap_max,
                             transparent=self.transparent, fq_layers=self.fq_layers, fq_dict_size=self.fq_dict_size,
                             attn_layers=self.attn_layers, no_const=self.no_const, rank=self.device,
                             classifier_labels=self.num_classes, capturable=self.graphs,
                             conditional=self.new_architecture, *args, **kwargs)
        self._graph_cache, self._static, self._graph_warm = {}, {}, set()  # graphs of a previous model instance are void
        if self.is_ddp:
            m = self.StylEx
            parallel.broadcast_parameters(m)
            self._d_sync = parallel.GradSync(list(m.D.parameters()))
            self._g_sync = parallel.GradSync(list(m.G.parameters()) + list(m.S.parameters())
                                             + list(m.encoder.parameters()))
            if self.device_rng and self.device.type == "cuda":
                # device-side draws: every rank gets its own latent / noise stream (cli.set_seed seeds all ranks alike,
                # like the reference's cli.py:49 — with the CPU draws that makes every rank see the same z)
                torch.cuda.manual_seed(torch.cuda.initial_seed() + 7919 * self.rank)

    def write_config(self):
        self.config_path.write_text(json.dumps(self.config()))

    def load_config(self):
        config = self.config() if not self.config_path.exists() else json.loads(self.config_path.read_text())
        self.image_size = config["image_size"]
        self.network_capacity = config["network_capacity"]
        self.transparent = config["transparent"]
        self.fq_layers = config["fq_layers"]
        self.fq_dict_size = config["fq_dict_size"]
        self.fmap_max = config.pop("fmap_max", 512)
        self.attn_layers = config.pop("attn_layers", [])
        self.no_const = config.pop("no_const", False)
        self.lr_mlp = config.pop("lr_mlp", 0.1)
        self.StylEx = None
        self.init_StylEx()

    def config(self):
        return {"image_size": self.image_size, "network_capacity": self.network_capacity, "lr_mlp": self.lr_mlp,
                "transparent": self.transparent, "fq_layers": self.fq_layers, "fq_dict_size": self.fq_dict_size,
                "attn_layers": self.attn_layers, "no_const": self.no_const,
                # the reference reads 'fmap_max' back (:1209) but never writes it (:1215-1218); writing it
                # keeps non-default widths loadable and stays readable by the reference
                "fmap_max": self.fmap_max}

    def set_data_src(self, folder="./", dataset_name=None):
        if dataset_name == "MNIST":
            self.dataset = MNIST_1vA(digit=8)
        num_workers = default(self.num_workers, NUM_CORES if not self.is_ddp else 0)
        if self.device_pipeline:
            import input_pipeline

            # the decode-only dataset has no augmentation stage (the reference's RandomApply(aug_prob, crop + flip) runs
            # on PIL images in the worker): refuse instead of silently training without it
            assert not self.dataset_aug_prob, "device_pipeline=True does not implement dataset_aug_prob > 0"
            ds = input_pipeline.RawImageFolder(folder, self.image_size, transparent=self.transparent)
            sampler = DistributedSampler(ds, rank=self.rank, num_replicas=self.world_size,
                                         shuffle=True) if self.is_ddp else None
            self.loader, self.dataset = input_pipeline.make_device_loader(
                ds, self.image_size, math.ceil(self.batch_size / self.world_size), self.device,
                num_workers=num_workers, transparent=self.transparent, sampler=sampler, shuffle=not self.is_ddp)
        else:
            self.dataset = Dataset(folder, self.image_size, transparent=self.transparent, aug_prob=self.dataset_aug_prob)
            sampler = DistributedSampler(self.dataset, rank=self.rank, num_replicas=self.world_size,
                                         shuffle=True) if self.is_ddp else None
            loader = data.DataLoader(self.dataset, num_workers=num_workers,
                                     batch_size=math.ceil(self.batch_size / self.world_size), sampler=sampler,
                                     shuffle=not self.is_ddp, drop_last=True, pin_memory=torch.cuda.is_available())
            self.loader = cycle(loader)
        num_samples = len(self.dataset)
        if not exists(self.aug_prob) and num_samples < 1e5:
            self.aug_prob = min(0.5, (1e5 - num_samples) * 3e-6)
            print(f"autosetting augmentation probability to {round(self.aug_prob * 100)}%")

    # ---- the hot path -------------------------------------------------------------------

    def _next_batch(self):
        batch = next(self.loader)
        if batch.device == self.device or self.device.type != "cuda" or _capturing():
            return batch.to(self.device, non_blocking=True)
        # loader batches (pinned by the DataLoader) take the upload stream too, see _Staging.upload
        cur = torch.cuda.current_stream(self.device)
        up = _Staging._upload_streams.get(self.device)
        if up is None:
            up = _Staging._upload_streams[self.device] = torch.cuda.Stream(self.device)
        with torch.cuda.stream(up):
            out = batch.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(up)
        cur.wait_event(ev)
        out.record_stream(cur)
        return out

    def _resolve_losses(self, raise_nan=True):
        """Wait for the pending loss copy of the last step (only that copy, not the stream) and publish the
        scalars.  NaN handling as the reference (:1483-1486): reload the last checkpoint and raise; train()
        calls this for the previous step once the next discriminator phase is enqueued, and before saving."""
        pend, self._pending = self._pending, None
        if pend is not None:
            host, done, has_gp, keep_rec = pend
            if done is not None:
                done.synchronize()
            vals = host.tolist()
            self.d_loss, self.g_loss = vals[0], vals[1]
            if has_gp:
                self.last_gp_loss = vals[4]
            if keep_rec:
                self.total_rec_loss, self.total_kl_loss = vals[2], vals[3]
            self._nan = vals[5] > 0  # isnan(d_loss) | isnan(g_loss), OR-ed over the ranks under DDP
        if raise_nan and getattr(self, "_nan", False):
            self._nan = False
            print(f"NaN detected for generator or discriminator. Loading from checkpoint #{self.checkpoint_num}")
            self.load(self.checkpoint_num)
            raise NanException

    def _fork(self, fns):
        """Evaluate independent sub-graphs concurrently: fns[0] on the current HIP stream, the others on side
        streams, joined before returning.  The small layers of D / encoder / classifier / LPIPS each fill a
        fraction of the 256 CUs; issued from separate streams they overlap, forward and — because autograd
        replays every node on its forward stream — backward.  Results are identical (no shared mutable state)."""
        if self.device.type != "cuda" or len(fns) < 2 or os.environ.get("STYLEX_STREAMS", "1") == "0":
            return [f() for f in fns]
        if getattr(self, "_side_streams", None) is None:
            self._side_streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
            # parameters shared by branches on different streams (encoder, classifier input) accumulate across
            # streams by design; the engine synchronises them — silence its advisory
            warn_off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if warn_off is not None:
                warn_off(False)
        main = torch.cuda.current_stream()
        outs, used = [None] * len(fns), []
        for i in range(1, len(fns)):
            side = self._side_streams[(i - 1) % len(self._side_streams)]
            if side not in used:
                side.wait_stream(main)  # everything the branch reads was enqueued on `main` before the fork
                used.append(side)
            with torch.cuda.stream(side):
                outs[i] = fns[i]()
            for t in (outs[i] if isinstance(outs[i], (tuple, list)) else (outs[i],)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)  # produced on a side stream, consumed (and freed) on `main`
        outs[0] = fns[0]()
        for side in used:
            main.wait_stream(side)
        return outs

    def _classify(self, images):
        if images.is_cuda and getattr(self.classifier, "accepts_any_layout", False) and os.environ.get("STYLEX_RESIZE_FUSE", "1") != "0":
            return self.classifier.classify_images(images)  # the fused resize kernel reads any layout: no dense copy
        return self.classifier.classify_images(_frozen_layout(images))

    def _styles_from_encoder(self, batch):
        """W of an encoder micro-step: encoder output ++ classifier logits (:1309-1314) — ++ classifier PROBABILITIES in
        the new architecture (stylex_train_new.py:1339-1343)."""
        m = self.StylEx
        enc, logits = self._fork([lambda: m.encoder(batch), lambda: self._classify(batch)])
        cond = F.softmax(logits, dim=1) if self.new_architecture else logits
        w = styles_def_to_tensor([(torch.cat((enc, cond), dim=1), m.G.num_layers)])
        return enc, logits, w

    def _styles_of(self, entry, probs=None):
        """[B, L, latent] style tensor of a noise micro-step: ('noise', [(z, n_layers), ...], inoise) as drawn by
        noise_list / mixed_list, or the graph form ('noise_static', (z1, z2, tt), inoise, None) whose layer split `tt` is a
        device scalar, so that one captured graph serves every split (pure selection: identical values)."""
        m = self.StylEx
        if entry[0] == "noise":
            return styles_def_to_tensor(latent_to_w(m.S, entry[1], probs))
        z1, z2, tt = entry[1]
        w1, w2 = m.S(z1), m.S(z2)
        first = torch.arange(m.G.num_layers, device=z1.device) < tt
        return torch.where(first[None, :, None], w1[:, None, :], w2[:, None, :])

    # -- host side of a phase: every RNG draw / loader fetch of its micro-steps, in the reference's order ---------

    def _draw_d(self, group, st, fuse):
        """Discriminator-phase inputs of one micro-step group (:1299-1333): per micro-step the real batch, then either
        (encoder micro-step) a second loader batch + image noise, or (noise micro-step) random() for mixed_prob,
        [torch.rand(()) split], 1-2 latents, image noise; then the two AugWrapper draws."""
        m = self.StylEx
        batch_size = math.ceil(self.batch_size / self.world_size)
        reals, micro = [], []
        new = self.new_architecture
        z_dim = m.G.latent_dim - (m.num_classes if new else 0)
        for _ in group:
            reals.append(self._next_batch())
            # new architecture: EVERY micro-step draws a conditioning batch (stylex_train_new.py:1329-1334); an
            # encoder micro-step encodes that same batch
            cond = self._next_batch() if new else None
            if (not self.alternating_training) or st["encoder_input"]:
                batch2 = cond if new else self._next_batch()
                micro.append(("enc", batch2, image_noise(batch_size, m.G.image_size, device=self.device), batch2))
                st["encoder_input"] = False
            else:
                st["latents_fn"] = mixed_list if random() < self.mixed_prob else noise_list
                style = st["latents_fn"](batch_size, m.G.num_layers, z_dim, device=self.device)
                micro.append(("noise", style, image_noise(batch_size, m.G.image_size, device=self.device), cond))
                if self.alternating_training:
                    st["encoder_input"] = True
            if fuse:
                random(), random()  # the two AugWrapper draws of this micro-step (:1331-1333)
        return reals, micro

    def _draw_g(self, group, st, fuse, apply_pl):
        """Generator-phase inputs of one group (:1376-1421): loader batch (drawn on noise micro-steps too), image
        noise / latents, the AugWrapper draw, and the path-length noise where the reference draws it."""
        m = self.StylEx
        batch_size = math.ceil(self.batch_size / self.world_size)
        micro, pl_noises = [], []
        for _ in group:
            batch = self._next_batch()
            if (not self.alternating_training) or st["encoder_input"]:
                micro.append(("enc", batch, image_noise(batch_size, m.G.image_size, device=self.device), batch))
            else:
                z_dim = m.G.latent_dim - (m.num_classes if self.new_architecture else 0)
                style = st["latents_fn"](batch_size, m.G.num_layers, z_dim, device=self.device)
                micro.append(("noise", style, image_noise(batch_size, m.G.image_size, device=self.device), batch))
            if fuse:
                random()  # the AugWrapper draw of this micro-step (:1417)
                if apply_pl:
                    pl_noises.append(draw_pl_noise((batch_size, 4 if self.transparent else 3, m.G.image_size,
                                                    m.G.image_size), self.device))
            st["encoder_input"] = not st["encoder_input"]
        return micro, pl_noises

    # -- device side of a phase: forward + backward of one micro-step group -------------------------------------

    def _d_compute(self, reals, micro, apply_gp, gae, fuse, last, acc):
        m = self.StylEx
        aug = {"prob": self.aug_prob, "types": self.aug_types}

        new = self.new_architecture

        def D_call(images, detach=False, probs=None):
            if fuse:  # AugWrapper at prob 0 is D itself; its random() draw is issued where the reference draws it
                images = images.detach() if detach else images
                return m.D(images, probabilities=probs) if new else m.D(images)
            return m.D_aug(images, detach=detach, probabilities=probs, **aug)

        cond = None
        with torch.no_grad():  # the generator/encoder graph is never used in this phase (:1330-1331)
            ws, conds = [], []
            for e in micro:
                if e[0] == "enc":
                    _, logits, w = self._styles_from_encoder(e[1])
                    p_i = F.softmax(logits, dim=1) if new else None
                else:
                    # new architecture: probabilities of the micro-step's conditioning batch (:1332-1333, :1352)
                    p_i = F.softmax(self._classify(e[3]), dim=1) if new else None
                    w = self._styles_of(e, p_i)
                ws.append(w)
                conds.append(p_i)
            if new:
                cond = _cat(conds)
            ops.set_fast(True)
            generated = m.G(_cat(ws), _cat([e[2] for e in micro]))
        real = _cat(reals)
        n_fake = generated.shape[0]
        grad_norms = None
        tangent_gp = False
        if apply_gp:
            import gp_tangent

            # default architecture, bf16 mode: D(real) and ||dD/dx|| as ONE first-order node (the penalty's parameter
            # gradient through a tangent pass instead of a double backward, gp_tangent.py)
            tangent_gp = fuse and not new and gp_tangent.supported(m.D, real)
            real = real.detach() if tangent_gp else real.detach().requires_grad_()

            def fake_branch():
                ops.set_fast(True)  # the fake branch is only ever differentiated once
                return D_call(generated, detach=True, probs=cond)

            def real_branch():
                if tangent_gp:
                    ops.set_fast(True)
                    return gp_tangent.d_real_with_norms(m.D, real)
                ops.set_fast(False)  # the gradient penalty differentiates the real branch twice
                return D_call(real, probs=cond)

            # the two D passes of a penalty step are independent until the loss: two HIP streams when D_aug is
            # a pass-through (`fuse`; with augmentation the reference's fake-then-real draw order is kept)
            if fuse:
                real_out, fake_out = self._fork([real_branch, fake_branch])
            else:
                fake_out, real_out = fake_branch(), real_branch()
            ops.set_fast(False)
            if tangent_gp:
                real_out, grad_norms = real_out
            else:
                grad_norms = gradient_norms(real, real_out)
        else:
            # D(fake) and D(real) are one pass over the concatenated batch
            ops.set_fast(True)
            if fuse:
                both = D_call(torch.cat((generated, real), dim=0), detach=True,
                              probs=torch.cat((cond, cond), dim=0) if new else None)
                fake_out, real_out = both[:n_fake], both[n_fake:]
            else:
                fake_out = D_call(generated, detach=True, probs=cond)
                real_out = D_call(real, probs=cond)
        disc_loss, lo = 0, 0
        for r in reals:
            sl = slice(lo, lo + r.shape[0])
            lo += r.shape[0]
            divergence = hinge_loss(real_out[sl], fake_out[sl])
            disc_loss = disc_loss + divergence
            if apply_gp:
                gp = 10 * ((grad_norms[sl] - 1) ** 2).mean()
                acc["gp"] = gp.detach()
                disc_loss = disc_loss + gp
            acc["d"] += divergence.detach() / gae
        if self.is_ddp and last and not _capturing():
            self._d_sync.arm()  # the all-reduce of D's gradients starts inside this backward
        disc_loss = disc_loss / gae
        if self._nan_hook:
            disc_loss.register_hook(raise_if_nan)  # reference :1352 (opt-in: the check synchronises the host)
        disc_loss.backward()

    def _g_compute(self, micro, pl_noises, apply_pl, gae, fuse, last, acc):
        m = self.StylEx
        aug = {"prob": self.aug_prob, "types": self.aug_types}

        new = self.new_architecture

        def D_call(images, probs=None):
            if fuse:
                return m.D(images, probabilities=probs) if new else m.D(images)
            return m.D_aug(images, detach=False, probabilities=probs, **aug)

        ws, encs, conds = [], [], []
        for e in micro:
            if e[0] == "enc":
                enc_out, real_logits, w_styles = self._styles_from_encoder(e[1])
                encs.append((e[1], enc_out, real_logits))
                p_i = F.softmax(real_logits, dim=1) if new else None
            else:
                # new architecture: the classifier runs on the loader batch of EVERY micro-step (:1437-1438)
                p_i = F.softmax(self._classify(e[3]), dim=1) if new else None
                w_styles = self._styles_of(e, p_i)
                encs.append(None)
            ws.append(w_styles)
            conds.append(p_i)
        cond = _cat(conds) if new else None
        w_all = _cat(ws)
        # path-length regularisation differentiates the GENERATOR twice (d images / d styles, then the loss on it):
        # only G runs on the composable double-differentiable ops; the encoder above and D / encoder / classifier /
        # LPIPS below are differentiated once and keep the fused path
        ops.set_fast(not apply_pl)
        generated_all = m.G(w_all, _cat([e[2] for e in micro]))
        ops.set_fast(True)
        pl_all = calc_pl_lengths(w_all, generated_all, _cat(pl_noises)) if (apply_pl and pl_noises) else None
        # four independent consumers of the generated batch: D, and per encoder micro-step the classifier,
        # the encoder and LPIPS — forked over HIP streams (see _fork)
        spans, lo = [], 0
        for w_styles in ws:
            spans.append(slice(lo, lo + w_styles.shape[0]))
            lo += w_styles.shape[0]
        branches, where = [lambda: D_call(generated_all, cond)], []
        for sl, enc in zip(spans, encs):
            if enc is not None:
                gen_i, batch_i = generated_all[sl], enc[0]
                where.append(len(branches))
                branches += [lambda g=gen_i: self._classify(g), lambda g=gen_i: m.encoder(g),
                             lambda g=gen_i, b=batch_i: perceptual_loss(b, g, self.lpips_fn)]
            else:
                where.append(None)
        outs = self._fork(branches)
        fake_all = outs[0]
        total_all = 0
        for sl, enc, at in zip(spans, encs, where):
            generated = generated_all[sl]
            loss = gen_hinge_loss(fake_all[sl], None)
            total = loss
            if apply_pl:
                pl_lengths = pl_all[sl] if pl_all is not None else calc_pl_lengths(w_all, generated)
                acc["pl"] = float(np.mean(pl_lengths.detach().cpu().numpy()))
                if not is_empty(self.pl_mean):
                    pl_loss = ((pl_lengths - self.pl_mean) ** 2).mean()
                    if not torch.isnan(pl_loss):
                        total = total + pl_loss
            total = total / gae
            if enc is not None:
                batch, enc_out, real_logits = enc
                gen_logits, gen_w, perceptual = outs[at:at + 3]
                twice = 1 if new else 2  # the new architecture doubled the scalings once at construction (:1166-1171)
                rec = twice * self.rec_scaling * reconstruction_loss(batch, generated, gen_w, enc_out,
                                                                     self.lpips_fn, perceptual=perceptual) / gae
                kl = twice * self.kl_scaling * classifier_kl_loss(real_logits, gen_logits) / gae
                total = total + rec + kl  # one backward == the three backward calls of :1436-1438
                acc["rec"] += rec.detach()
                acc["kl"] += kl.detach()
            total_all = total_all + total
            acc["g"] += loss.detach() / gae
        if self.is_ddp and last and not _capturing():
            self._g_sync.arm()
        if self._nan_hook:
            total_all.register_hook(raise_if_nan)  # reference :1352, :1432 (opt-in: it synchronises the host)
        total_all.backward()

    # -- draw-ahead worker -------------------------------------------------------------------------------------

    def _draw_submit(self, fn):
        if self._draw_worker is None:
            from concurrent.futures import ThreadPoolExecutor

            def init(device):
                if device.type == "cuda":
                    torch.cuda.set_device(device)
                _Staging._tls.device_rng = False

            self._draw_worker = ThreadPoolExecutor(1, thread_name_prefix="stylex-draw", initializer=init,
                                                   initargs=(self.device,))
        return self._draw_worker.submit(fn)

    def _drain_draw_ahead(self):
        """Wait for (and keep) a prefetched discriminator-phase draw: called before anything else draws from the loader
        or the CPU generators (evaluate, generate_truncated, calculate_fid ...)."""
        nd = getattr(self, "_next_d", None)
        if nd is not None:
            nd[0].exception()  # waits; an exception resurfaces when the next train() call takes the result

    def _step_ends_with_draws(self):
        """Does THIS train() call end with evaluate / save / FID (which consume loader batches and generator draws)?"""
        s = self.steps
        if s % self.save_every == 0 or s % self.evaluate_every == 0 or (s % 100 == 0 and s < 2500):
            return True
        return exists(self.calculate_fid_every) and s % self.calculate_fid_every == 0 and s != 0

    def _new_acc(self):
        acc = {k: torch.zeros((), device=self.device) for k in ("d", "g", "rec", "kl")}
        acc["gp"], acc["pl"] = None, self.pl_mean
        return acc

    def _d_phase(self, groups, inputs, apply_gp, gae, fuse, acc, st=None):
        """zero_grad + forward/backward of every group; `inputs` None = draw each group's inputs right before its
        compute (eager mode: the host prepares the next group while the GPU runs this one)."""
        self._zero_grad("D")
        for gi, group in enumerate(groups):
            reals, micro = inputs[gi] if inputs is not None else self._draw_d(group, st, fuse)
            self._d_compute(reals, micro, apply_gp, gae, fuse, group is groups[-1], acc)

    def _g_phase(self, groups, inputs, apply_pl, gae, fuse, acc, st=None):
        m = self.StylEx
        ops.set_fast(True)  # first-order everywhere except the generator of a path-length step (see _g_compute)
        self._zero_grad("G")
        set_requires_grad(m.D, False)  # D weight-gradients of this phase are discarded by :1297 anyway
        try:
            for gi, group in enumerate(groups):
                micro, pl_noises = inputs[gi] if inputs is not None else self._draw_g(group, st, fuse, apply_pl)
                self._g_compute(micro, pl_noises, apply_pl, gae, fuse, group is groups[-1], acc)
        finally:
            set_requires_grad(m.D, True)
            ops.set_fast(False)

    def _zero_grad(self, which):
        m = self.StylEx
        if self.is_ddp:  # gradients are views of the persistent flat buckets the collectives run on
            (self._d_sync if which == "D" else self._g_sync).zero_grad()
        else:
            (m.D_opt if which == "D" else m.G_opt).zero_grad()

    def _loss_stack(self, acc):
        gp = acc["gp"]
        return torch.stack((acc["d"], acc["g"], acc["rec"], acc["kl"], gp if gp is not None else acc["d"]))

    @_own_rng_mode
    def train(self):
        """One optimiser step of D, then one of G (reference :1249-1506)."""
        assert exists(self.loader), "You must first initialize the data source with `.set_data_src(<folder of images>)`"
        if not exists(self.StylEx):
            self.init_StylEx()
        m = self.StylEx
        if not m.training:  # Module.train() walks ~500 submodules: only when evaluate() / a caller left eval mode on
            m.train()
        gae = self.gradient_accumulate_every
        apply_gp = self.steps % self.gp_every == 0
        apply_pl = (not self.no_pl_reg) and self.steps > self.pl_after and self.steps % self.pl_every == 0

        # Micro-step batching: G, D and the encoder carry no batch statistics, so the `gae` micro-steps of a
        # phase are evaluated as ONE pass over their concatenated batch (same per-sample arithmetic, the
        # weight-gradient GEMMs sum over gae*B rows at once, half the launches of the launch-bound small
        # layers).  Inputs are still DRAWN micro-step by micro-step in the reference's order (loader, Python
        # random(), torch.randn), so every RNG stream is consumed identically.  With augmentation on
        # (AugWrapper consumes RNG per call) every micro-step stays its own group.
        fuse = not self.aug_prob
        groups = [list(range(gae))] if fuse else [[i] for i in range(gae)]
        st = {"encoder_input": False, "latents_fn": None}

        self._calls = getattr(self, "_calls", 0) + 1
        if self._graphs_enabled() and fuse and not apply_pl and not self.new_architecture and self._calls > self.graph_warmup:
            acc_host = self._train_graphed(groups[0], st, apply_gp, gae)
            has_gp = apply_gp
        else:
            acc = self._new_acc()
            stale, self._g_fut = getattr(self, "_g_fut", None), None
            if stale is not None:  # a call that raised (NaN restart) left its generator-phase draw running: let it finish
                stale.exception()  # before anything else touches the loader / the generators
            # draw-ahead is an eager-path feature: with graphs enabled a draw prefetched here would be neither consumed
            # nor drained by a following _train_graphed call (which draws on its own), reordering the RNG / loader streams
            ahead = self._draw_mode if (fuse and not self.device_rng and not self._graphs_enabled()) else 0
            d_in = g_fut = None
            if ahead:
                sig = (gae, self.batch_size, self.world_size, self.alternating_training, self.new_architecture,
                       m.G.image_size, self.transparent)
                if self._next_d is not None:  # drawn under the previous call's generator phase
                    (fut, fsig), self._next_d = self._next_d, None
                    res = fut.result()
                    if fsig == sig:
                        d_in, st = res
                if d_in is None:
                    d_in = self._draw_d(groups[0], st, True)
                st_g = dict(st)

                def draw_g():
                    if self.alternating_training:
                        st_g["encoder_input"] = False
                    return self._draw_g(groups[0], st_g, True, apply_pl)

                g_fut = self._g_fut = self._draw_submit(draw_g)  # the generator phase's inputs, under the discriminator phase
            # ---------------- discriminator phase ----------------
            self._d_phase(groups, [d_in] if ahead else None, apply_gp, gae, fuse, acc, st)
            if self.is_ddp:
                self._d_sync.all_reduce()
            self._resolve_losses()  # previous step's scalars: its copy finished long ago, the GPU keeps running
            self._opt_step(m.D_opt)
            # ---------------- generator phase ----------------
            if self.alternating_training:
                st["encoder_input"] = False
            g_in = None
            if ahead:
                g_in = [g_fut.result()]
                self._g_fut = None
                if ahead > 1 and not self._step_ends_with_draws():
                    st_n = {"encoder_input": False, "latents_fn": None}
                    grp = list(groups[0])
                    self._next_d = (self._draw_submit(lambda: (self._draw_d(grp, st_n, True), st_n)), sig)
            self._g_phase(groups, g_in, apply_pl, gae, fuse, acc, st)
            if self.is_ddp:
                self._g_sync.all_reduce()
            self._opt_step(m.G_opt)
            acc_host, has_gp = self._loss_stack(acc), acc["gp"] is not None
            if apply_pl and not np.isnan(acc["pl"]):  # EMA(0.99), reference :1128, :1471-1473
                avg = float(acc["pl"])
                self.pl_mean = avg if self.pl_mean is None else self.pl_mean * 0.99 + 0.01 * avg

        # all loss scalars of the step leave in one asynchronous copy (resolved lazily, see _resolve_losses); under
        # DDP a NaN flag, OR-ed over the ranks ON THE DEVICE (no host sync), rides along so that every rank takes the
        # checkpoint-reload path at the same step
        stack = acc_host
        flag = torch.isnan(stack[:2]).any().to(stack.dtype).reshape(1)
        if self.is_ddp:
            parallel.all_reduce_max_(flag)
        stack = torch.cat((stack, flag))
        keep_rec = (not self.alternating_training) or gae > 1
        if stack.is_cuda:
            host = torch.empty(stack.shape, dtype=stack.dtype, pin_memory=True)
            host.copy_(stack, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        else:
            host, done = stack, None
        self._pending = (host, done, has_gp, keep_rec)
        if exists(self.tb_writer):
            for k, v in (("G", self.g_loss), ("D", self.d_loss), ("rec", self.total_rec_loss),
                         ("kl", self.total_kl_loss)):
                self.tb_writer.add_scalar("loss/" + k, v, self.steps)

        if self.is_main and self.steps % 10 == 0 and self.steps > 20000:
            m.EMA()
        if self.is_main and self.steps <= 25000 and self.steps % 1000 == 2:
            m.reset_parameter_averaging()
        if self.steps % self.save_every == 0 or self.steps % self.evaluate_every == 0:
            # never checkpoint a NaN state (reference :1483-1486 precedes the save).  On EVERY rank: the flag is already
            # MAX-reduced, and the reload it triggers broadcasts parameters — a rank that skipped this call would be in
            # the next phase's gradient all-reduce while rank 0 sits in that broadcast
            self._resolve_losses()
        if self.is_main:
            if self.steps % self.save_every == 0:
                self.save(self.checkpoint_num)
            if self.steps % self.evaluate_every == 0 or (self.steps % 100 == 0 and self.steps < 2500):
                self.evaluate(encoder_input=self.sample_from_encoder, num=floor(self.steps / self.evaluate_every))
            if exists(self.calculate_fid_every) and self.steps % self.calculate_fid_every == 0 and self.steps != 0:
                self.last_fid = self.calculate_fid(math.ceil(self.calculate_fid_num_images / self.batch_size))
        self.steps += 1
        self.av = None

    # -- HIP-graph replay of the whole step ---------------------------------------------------------------------

    def _graphs_enabled(self):
        """Whole-step HIP graphs: opt-in with Trainer(graphs=True) / STYLEX_GRAPHS=1 (bench.py turns them on).  A step
        is ~1100 kernel launches issued through ctypes/ATen (~70 ms of host time per step); captured once
        per step shape (with / without the gradient penalty) it replays with one hipGraphLaunch."""
        return self.graphs and self.device.type == "cuda"

    def _opt_step(self, opt):
        """Optimiser step + invalidation of the cached operand copies of the weights it changed (bf16 GEMM layouts,
        scaled mapping-network weights, ...).  The cache entries are valid for one modification stamp of their
        parameter; the fused Adam of the speed mode does NOT bump Parameter._version, so the stamp is set here
        (hb.mark_updated) — without it the generator phase evaluated D with its pre-update packs.
        Parameters another optimiser owns keep their packs (D's survive the generator's step).
        STYLEX_PREPACK=1 additionally rebuilds the invalidated copies right away on a side stream (hb.prepack) —
        measured slower than packing at first use (the ~60 tiny launches then compete with the start of the next
        forward pass instead of hiding under it); kept as an opt-in."""
        if self.device.type != "cuda":
            opt.step()
            return
        import hip_backend as hb

        params = [p for g in opt.param_groups for p in g["params"]]
        prepack = os.environ.get("STYLEX_PREPACK", "0") == "1"
        if prepack:
            hb.prepack_join()  # earlier prepacks have read the parameters
        # bf16 speed mode: Adam + the refresh of every cached operand copy of the stepped weights in ONE launch
        # (hb.adam_pack_step / csrc/adam_pack.hip: it stamps the parameters and re-validates the copies it rewrote);
        # STYLEX_ADAM_PACK=0, the first step of an optimiser, graphs and the fp32 parity mode take torch's own step
        if not (ops.get_precision() == "bf16" and not _capturing() and hb.adam_pack_step(opt)):
            opt.step()
            # the fused Adam does not bump Parameter._version: stamp what it stepped, or every cached operand copy of
            # these weights (bf16 GEMM layouts, scaled mapping-network weights, ...) would be served stale
            hb.mark_updated(params)
        if prepack:
            hb.prepack(params)

    def _bump_packs(self):
        import hip_backend as hb

        hb.pack_cache_clear()  # optimiser steps inside a replayed graph do not advance Parameter._version

    def _bind(self, key, t):
        """Copy a freshly drawn input into the static device buffer the captured graph reads."""
        buf = self._static.get(key)
        if buf is None:
            buf = self._static[key] = torch.empty_like(t)
        buf.copy_(t, non_blocking=True)
        return buf

    def _bind_micro(self, phase, i, e, layers):
        if e[0] == "enc":
            x = self._bind((phase, i, "x"), e[1])
            return ("enc", x, self._bind((phase, i, "n"), e[2]), x)
        style = e[1]
        z1 = self._bind((phase, i, "z1"), style[0][0])
        z2 = self._bind((phase, i, "z2"), style[1][0] if len(style) > 1 else style[0][0])
        tt = self._static.get((phase, i, "tt"))
        if tt is None:
            tt = self._static[(phase, i, "tt")] = torch.zeros((), dtype=torch.int64, device=self.device)
        tt.fill_(style[0][1] if len(style) > 1 else layers)
        return ("noise_static", (z1, z2, tt), self._bind((phase, i, "n"), e[2]), None)

    def _train_graphed(self, group, st, apply_gp, gae):
        m = self.StylEx
        layers = m.G.num_layers
        # host: every draw of the step in the reference's order, then into the static input buffers.  The previous
        # replay is still running on the GPU while this happens (stream-ordered copies).
        reals, micro_d = self._draw_d(group, st, True)
        if self.alternating_training:
            st["encoder_input"] = False
        micro_g, _ = self._draw_g(group, st, True, False)
        reals = [self._bind(("d", i, "real"), r) for i, r in enumerate(reals)]
        micro_d = [self._bind_micro("d", i, e, layers) for i, e in enumerate(micro_d)]
        micro_g = [self._bind_micro("g", i, e, layers) for i, e in enumerate(micro_g)]
        entry = self._graph_cache.get(apply_gp)
        if entry is None:
            self._resolve_losses()
        if entry is None and apply_gp not in self._graph_warm:
            # first eligible call of this step shape: run the exact code path of the capture (static input buffers,
            # device-side layer split) EAGERLY once — every kernel it launches must have been loaded before a capture
            # starts (a first-time kernel load inside a capture is not capturable)
            self._graph_warm.add(apply_gp)
            acc = self._new_acc()
            self._d_phase([group], [(reals, micro_d)], apply_gp, gae, True, acc)
            if self.is_ddp:
                self._d_sync.all_reduce()
            self._opt_step(m.D_opt)
            self._g_phase([group], [(micro_g, [])], False, gae, True, acc)
            if self.is_ddp:
                self._g_sync.all_reduce()
            self._opt_step(m.G_opt)
            self._bump_packs()
            return self._loss_stack(acc)
        if entry is None:
            entry = self._capture(apply_gp, gae, [group], [(reals, micro_d)], [(micro_g, [])])
            self._graph_cache[apply_gp] = entry
        resolve = self._pending is not None
        graphs, out = entry
        syncs = [self._d_sync.all_reduce, self._g_sync.all_reduce, None] if self.is_ddp else [None, None, None]
        for gi, (g, sync) in enumerate(zip(graphs, syncs)):
            g.replay()
            if gi == 0 and resolve:
                # previous step's scalars, read where the eager path reads them: AFTER this step's discriminator phase
                # is queued, so the GPU never runs dry while the host waits for that copy
                self._resolve_losses()
            if sync is not None:
                sync()
        self._bump_packs()
        return out

    def _capture(self, apply_gp, gae, groups, d_in, g_in):
        """Capture the step as three HIP graphs (D forward/backward | D step + G forward/backward | G step); under
        DDP the RCCL gradient all-reduces are issued between the replays."""
        m = self.StylEx
        acc = {}

        def seg_d():
            acc.update(self._new_acc())
            self._d_phase(groups, d_in, apply_gp, gae, True, acc)
            if self.is_ddp:
                self._d_sync.pack_all()  # captured: every replay refills the flat buckets the collectives run on

        def seg_g():
            self._opt_step(m.D_opt)  # + stamp: the generator phase must re-pack D's updated weights inside the capture
            self._g_phase(groups, g_in, False, gae, True, acc)
            if self.is_ddp:
                self._g_sync.pack_all()

        def seg_tail():
            self._opt_step(m.G_opt)
            acc["out"] = self._loss_stack(acc)

        # always three graphs: each phase captures fine on its own, but D phase + G phase in ONE capture with the
        # branch streams on crashes hipStreamEndCapture (ROCm 7.2, tools/graph_stage_probe.py stage "step")
        segments = [[seg_d], [seg_g], [seg_tail]]
        torch.cuda.synchronize()
        dbg = os.environ.get("STYLEX_GRAPH_DEBUG", "0") == "1"
        share_pool = os.environ.get("STYLEX_GRAPH_POOL", "1") == "1"
        mode = os.environ.get("STYLEX_GRAPH_MODE", "thread_local")
        if self._graph_pool is None and share_pool:
            self._graph_pool = torch.cuda.graph_pool_handle()
        graphs = []
        for si, fns in enumerate(segments):
            self._bump_packs()
            g = torch.cuda.CUDAGraph()
            if dbg:
                print("capture: gp=%s segment %d begin" % (apply_gp, si), flush=True)
            with torch.cuda.graph(g, pool=self._graph_pool if share_pool else None, capture_error_mode=mode):
                for fn in fns:
                    fn()
                if dbg:
                    print("capture: segment %d body recorded, ending capture" % si, flush=True)
            if dbg:
                print("capture: segment %d done" % si, flush=True)
            graphs.append(g)
        self._bump_packs()
        return graphs, acc["out"]

    # ---- evaluation / generation (reference :1508-1698) ---------------------------------------

    @torch.no_grad()
    @_own_rng_mode
    def evaluate(self, encoder_input=False, num=0, trunc=1.0):
        m = self.StylEx
        m.eval()
        rows = self.num_image_tiles
        latent_dim, image_size, num_layers = m.G.latent_dim, m.G.image_size, m.G.num_layers
        new = self.new_architecture
        z_dim = latent_dim - (m.num_classes if new else 0)
        latents = noise_list(rows ** 2, num_layers, z_dim, device=self.device)
        n = image_noise(rows ** 2, image_size, device=self.device)
        tag = ""
        # default architecture: a loader batch is always drawn (:1526); the new one draws it only with encoder_input
        # (stylex_train_new.py:1602-1607) and then has nothing to put next to the samples (it crashes there; here the
        # grid simply holds the samples)
        image_batch = self._next_batch() if (encoder_input or not new) else None
        w, probs = None, None
        if encoder_input:
            tag = "from_encoder"
            logits = self.classifier.classify_images(image_batch)
            if new:
                probs = F.softmax(logits, dim=1)
                w = [(m.encoder(image_batch), num_layers)]  # probabilities are appended in generate_truncated
            else:
                w = [(torch.cat((m.encoder(image_batch), logits), dim=1), num_layers)]
            rows = len(image_batch)
        elif new:  # random class probabilities (stylex_train_new.py:1618-1620)
            probs = torch.rand(rows ** 2, 2, device=self.device)
            probs = probs / torch.sum(probs, dim=1, keepdim=True)
        out_dir = self.results_dir / self.name

        def grid(imgs, path):
            save_image_grid(imgs if image_batch is None else torch.cat((image_batch, imgs)), path, nrow=rows)

        imgs = self.generate_truncated(m.S, m.G, latents, n, w=w, trunc_psi=self.trunc_psi, probabilities=probs)
        grid(imgs, str(out_dir / f"{num}-{tag}.png"))
        imgs = self.generate_truncated(m.SE, m.GE, latents, n, w=w, trunc_psi=self.trunc_psi, probabilities=probs)
        grid(imgs, str(out_dir / f"{num}-{tag}-ema.png"))
        nn_ = noise(rows, z_dim, device=self.device)
        tiled = nn_.repeat_interleave(rows, dim=0)
        repeated = nn_.repeat(rows, 1)
        tt = int(num_layers / 2)
        mixed = [(tiled, tt), (repeated, num_layers - tt)]
        if new:
            probs = torch.rand(rows ** 2, 2, device=self.device)
            probs = probs / torch.sum(probs, dim=1, keepdim=True)
        imgs = self.generate_truncated(m.SE, m.GE, mixed, n, trunc_psi=self.trunc_psi, probabilities=probs)
        grid(imgs, str(out_dir / f"{num}-{tag}-mr.png"))

    @torch.no_grad()
    def calculate_fid(self, num_batches):
        raise RuntimeError("FID needs pytorch_fid + Inception weights (network) — out of scope offline")

    @torch.no_grad()
    @_own_rng_mode
    def truncate_style(self, tensor, trunc_psi=0.75):
        m = self.StylEx
        if not exists(self.av):
            z = noise(2000, m.G.latent_dim - (m.num_classes if self.new_architecture else 0), device=self.device)
            samples = evaluate_in_chunks(self.batch_size, m.S, z).cpu().numpy()
            self.av = np.expand_dims(np.mean(samples, axis=0), axis=0)
        av = torch.from_numpy(self.av).to(self.device)
        return trunc_psi * (tensor - av) + av

    @torch.no_grad()
    def truncate_style_defs(self, w, trunc_psi=0.75):
        return [(self.truncate_style(t, trunc_psi=trunc_psi), n) for t, n in w]

    @torch.no_grad()
    def generate_truncated(self, S, G, style, noi, w=None, trunc_psi=0.75, num_image_tiles=8, probabilities=None):
        if w is None:
            w = [(S(z), n) for z, n in style]
        w_truncated = self.truncate_style_defs(w, trunc_psi=trunc_psi)
        if self.new_architecture:
            # stylex_train_new.py:1741-1745: only the FIRST truncated latent is used (a mixed pair loses its second
            # half there) with the class probabilities appended, for all layers
            w_truncated = [(torch.cat((w_truncated[0][0], probabilities), dim=1), self.StylEx.G.num_layers)]
        w_styles = styles_def_to_tensor(w_truncated)
        # with encoder input the styles have len(image_batch) rows while the noise was drawn for num_image_tiles^2: the
        # reference's chunk-wise zip pairs row i with row i and drops the rest (:1653) — which only works while both
        # split into equally sized chunks (under DDP the loader batch is batch_size / world rows, and the first chunk
        # of styles no longer matches the first chunk of noise); pairing row by row explicitly is the same result
        noi = noi[:w_styles.shape[0]]
        return evaluate_in_chunks(self.batch_size, G, w_styles, noi).clamp_(0., 1.)

    @torch.no_grad()
    @_own_rng_mode
    def generate_interpolation(self, num=0, num_image_tiles=8, trunc=1.0, num_steps=100, save_frames=False):
        m = self.StylEx
        m.eval()
        rows = num_image_tiles
        latent_dim, image_size, num_layers = m.G.latent_dim, m.G.image_size, m.G.num_layers
        low = noise(rows ** 2, latent_dim, device=self.device)
        high = noise(rows ** 2, latent_dim, device=self.device)
        n = image_noise(rows ** 2, image_size, device=self.device)
        frames = []
        for ratio in torch.linspace(0., 8., num_steps):
            latents = [(slerp(ratio, low, high), num_layers)]
            imgs = self.generate_truncated(m.SE, m.GE, latents, n, trunc_psi=self.trunc_psi)
            frames.append(grid_to_pil(imgs, nrow=rows))
        frames[0].save(str(self.results_dir / self.name / f"{num}.gif"), save_all=True, append_images=frames[1:],
                       duration=80, loop=0, optimize=True)
        if save_frames:
            folder = self.results_dir / self.name / f"{num}"
            folder.mkdir(parents=True, exist_ok=True)
            for i, fr in enumerate(frames):
                fr.save(str(folder / f"{i}.{self.image_extension}"))

    def print_log(self):
        data_ = [("G", self.g_loss), ("D", self.d_loss), ("GP", self.last_gp_loss), ("PL", self.pl_mean),
                 ("CR", self.last_cr_loss), ("Q", self.q_loss), ("FID", self.last_fid), ("Rec", self.total_rec_loss),
                 ("KL", self.total_kl_loss)]
        print(" | ".join(f"{k}: {v:.2f}" for k, v in data_ if exists(v)))

    def track(self, value, name):
        pass  # aim logger is not available offline (reference :1717-1720)

    def model_name(self, num):
        return str(self.models_dir / self.name / f"model_{num}.pt")

    def init_folders(self):
        (self.results_dir / self.name).mkdir(parents=True, exist_ok=True)
        (self.models_dir / self.name).mkdir(parents=True, exist_ok=True)

    def clear(self):
        rmtree(str(self.models_dir / self.name), True)
        rmtree(str(self.results_dir / self.name), True)
        rmtree(str(self.fid_dir), True)
        rmtree(str(self.config_path), True)
        self.init_folders()

    def save(self, num):
        if not exists(self.StylEx):
            self.init_StylEx()
        data = {"StylEx": self.StylEx.state_dict(), "version": __version__}
        if self.save_training_state:
            # extension: the reference drops optimiser moments, step count and the path-length mean on
            # every restart; extra keys are ignored by the reference's loader (:1707-1716)
            data["training_state"] = {"G_opt": self.StylEx.G_opt.state_dict(), "D_opt": self.StylEx.D_opt.state_dict(),
                                      "steps": self.steps,
                                      "pl_mean": None if self.pl_mean is None else float(self.pl_mean)}
        torch.save(data, self.model_name(num))
        self.write_config()

    def load(self, num=-1):
        self.load_config()
        name = num
        if num == -1:
            saved = sorted(int(p.stem.split("_")[1]) for p in Path(self.models_dir / self.name).glob("model_*.pt"))
            if not saved:
                return
            name = saved[-1]
            print(f"continuing from previous epoch - {name}")
        self.steps = name * self.save_every
        load_data = torch.load(self.model_name(name), map_location=self.device)
        if "version" in load_data:
            print(f"loading from version {load_data['version']}")
        self.StylEx.load_state_dict(load_data["StylEx"])
        extra = load_data.get("training_state")
        if extra is not None and self.save_training_state:
            self.StylEx.G_opt.load_state_dict(extra["G_opt"])
            self.StylEx.D_opt.load_state_dict(extra["D_opt"])
            self.steps, self.pl_mean = extra["steps"], extra["pl_mean"]
        if self.is_ddp:
            parallel.broadcast_parameters(self.StylEx)
        import hip_backend as hb

        hb.mark_updated(self.StylEx.parameters())  # belt and braces: no operand pack of the pre-load weights survives
        hb.adam_forget()  # load_state_dict installed new moment / step tensors: no fused-Adam plan keeps the old pointers


def grid_to_pil(images, nrow=8, padding=2):
    """torchvision.utils.make_grid + ToPILImage equivalent (torchvision is absent offline)."""
    from PIL import Image

    images = images.detach().float().cpu().clamp(0, 1)
    n, c, h, w = images.shape
    ncol = min(nrow, n)
    nrows = int(math.ceil(n / ncol))
    grid = torch.zeros(c, nrows * (h + padding) + padding, ncol * (w + padding) + padding)
    for i in range(n):
        r, q = divmod(i, ncol)
        y0, x0 = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = images[i]
    arr = (grid.permute(1, 2, 0).numpy() * 255 + 0.5).astype(np.uint8)
    return Image.fromarray(arr if c != 1 else arr[:, :, 0])


def save_image_grid(images, path, nrow=8):
    grid_to_pil(images, nrow=nrow).save(path)


class ModelLoader:  # reference :1777-1800
    def __init__(self, *, base_dir, name="default", load_from=-1, **trainer_kwargs):
        trainer_kwargs.setdefault("classifier_name", "resnet")
        self.model = Trainer(name=name, base_dir=base_dir, **trainer_kwargs)
        self.model.load(load_from)

    def noise_to_styles(self, noise_, trunc_psi=None):
        w = self.model.StylEx.SE(noise_.to(self.model.device))
        if exists(trunc_psi):
            w = self.model.truncate_style(w)
        return w

    def styles_to_images(self, w):
        m = self.model.StylEx
        w_tensors = styles_def_to_tensor([(w, m.GE.num_layers)])
        images = m.GE(w_tensors, image_noise(w.shape[0], self.model.image_size, device=self.model.device))
        return images.clamp_(0., 1.)
