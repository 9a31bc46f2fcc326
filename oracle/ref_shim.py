"""TEST INFRASTRUCTURE — runs ONLY in the build container (needs /root/reference).

Imports the reference's own ``stylex/stylex_train.py`` on CPU by registering
stub modules for the third-party packages that are absent offline, so that the
reference itself can generate golden vectors (``oracle/make_golden.py``) and
validate the restatement in ``oracle/stylex_oracle.py``.

Nothing from here travels to the GPU box as an executable dependency: the
``-m gpu`` tests, ``smoke()`` and ``bench.py`` never import this file (the seeded
stand-in classifiers they need live in ``oracle/standins.py``, which has no
reference import and no monkey-patch in it).

Third-party semantics restated here (each one is "parity unpinned" at the
package boundary because the real package is not installable offline):

* kornia==0.6.2 ``filter2d(x, k, border_type='reflect', normalized=True)``
  (call site stylex_train.py:153) = reflect-pad by k//2 then depthwise
  correlation with k / sum|k|.
* torchvision==0.11.1 ``transforms.functional.resize`` on tensors
  (resnet_classifier.py:61) = bilinear, align_corners=False, no antialias.
* torchvision ``transforms.Normalize`` = (x - mean) / std per channel.
* lpips==0.1.4 ``LPIPS(net='alex')``: AlexNet features + learned 1x1 heads;
  the pretrained weights are not available offline, so a seeded
  random-weight network of the published architecture is used
  (``oracle/lpips_standin.py``) — fixtures record that fact.
"""
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("STYLEX_REFERENCE", "/root/reference")
REF_STYLEX = os.path.join(REF_ROOT, "stylex")

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)


def reference_available():
    return os.path.isfile(os.path.join(REF_STYLEX, "stylex_train.py"))


def _filter2d(x, kernel, border_type="reflect", normalized=False):
    k = kernel.to(x).unsqueeze(1)  # [1,1,kh,kw]
    if normalized:
        k = k / k.abs().sum(dim=(-2, -1), keepdim=True)
    c = x.shape[1]
    kh, kw = k.shape[-2:]
    k = k.expand(c, 1, kh, kw)
    xp = F.pad(x, [kw // 2, kw // 2, kh // 2, kh // 2], mode=border_type)
    return F.conv2d(xp, k, groups=c)


class _Identity:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


class _Compose:
    def __init__(self, ts):
        self.ts = list(ts)

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _Lambda:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x)


class _Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean).view(1, -1, 1, 1)
        self.std = torch.tensor(std).view(1, -1, 1, 1)

    def __call__(self, x):
        return (x - self.mean.to(x)) / self.std.to(x)


# ---- torchvision==0.11.1 transforms on PIL images, restated (package absent offline: "parity unpinned" at this
# boundary; the REFERENCE's use of them — order, arguments, RandomApply draw, mode conversion — is what the Dataset
# fixture pins).  Each function names the torchvision function it restates.

def _tv_resize(img, size):
    """torchvision.transforms.functional.resize: tensors -> F.interpolate(bilinear, align_corners=False, no antialias)
    (functional_tensor.resize, used by resnet_classifier.py:61); PIL images -> functional_pil.resize: an int size
    scales the SHORTER side to `size` and the longer one to int(size * long / short) (truncation), returns the image
    unchanged when the shorter side already equals `size`; a (h, w) pair resizes to exactly that.  PIL bilinear."""
    if isinstance(img, torch.Tensor):
        return F.interpolate(img, size=size, mode="bilinear", align_corners=False)
    from PIL import Image

    if isinstance(size, (list, tuple)) and len(size) == 1:
        size = size[0]
    if isinstance(size, int):
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        if short == size:
            return img
        new_short, new_long = size, int(size * long / short)
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
        return img.resize((new_w, new_h), Image.BILINEAR)
    return img.resize(tuple(size[::-1]), Image.BILINEAR)


class _Resize:
    def __init__(self, size, *a, **k):
        self.size = size

    def __call__(self, img):
        return _tv_resize(img, self.size)


class _CenterCrop:
    """transforms.CenterCrop(int) -> functional.center_crop: offsets int(round((side - crop) / 2.0)) — Python's
    round-half-to-even — after zero-padding an image smaller than the crop."""

    def __init__(self, size):
        self.size = (int(size), int(size)) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        from PIL import ImageOps

        ch, cw = self.size
        w, h = img.size
        if cw > w or ch > h:
            l, t = (cw - w) // 2 if cw > w else 0, (ch - h) // 2 if ch > h else 0
            r, b = (cw - w + 1) // 2 if cw > w else 0, (ch - h + 1) // 2 if ch > h else 0
            img = ImageOps.expand(img, border=(l, t, r, b), fill=0)
            w, h = img.size
            if (cw, ch) == (w, h):
                return img
        top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
        return img.crop((left, top, left + cw, top + ch))


class _RandomResizedCrop:
    """transforms.RandomResizedCrop(size, scale, ratio): get_params draws (torch global RNG) up to 10 x
    [uniform area fraction, log-uniform aspect ratio], takes the first box that fits with randint offsets, else the
    central fallback; forward = functional.resized_crop = crop, then resize to (size, size) bilinear."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):
        self.size = (int(size), int(size)) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio = scale, ratio

    @staticmethod
    def get_params(img, scale, ratio):
        import math

        width, height = img.size
        area = height * width
        log_ratio = torch.log(torch.tensor(ratio))
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, size=(1,)).item()
                j = torch.randint(0, width - w + 1, size=(1,)).item()
                return i, j, h, w
        in_ratio = float(width) / float(height)
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def __call__(self, img):
        i, j, h, w = self.get_params(img, self.scale, self.ratio)
        return _tv_resize(img.crop((j, i, j + w, i + h)), self.size)


class _ToTensor:
    """transforms.ToTensor -> functional.to_tensor for 8-bit PIL modes: HWC bytes -> CHW float32, .div(255)."""

    def __call__(self, pic):
        import numpy as np

        arr = np.array(pic, np.uint8, copy=True)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("fire", Fire=lambda f: None)
    from lpips_standin import LPIPSStandIn

    mod("lpips", LPIPS=LPIPSStandIn)
    mod("aim", Session=object)
    mod("vector_quantize_pytorch", VectorQuantize=object)
    mod("torch.utils.tensorboard", SummaryWriter=object)
    kf = mod("kornia.filters", filter2d=_filter2d)
    mod("kornia", filters=kf)

    tf = mod("torchvision.transforms.functional", resize=_tv_resize)
    tr_attrs = dict(Compose=_Compose, Lambda=_Lambda, Resize=_Resize, RandomResizedCrop=_RandomResizedCrop,
                    CenterCrop=_CenterCrop, ToTensor=_ToTensor, Normalize=_Normalize, ToPILImage=_Identity,
                    functional=tf)
    tt = mod("torchvision.transforms", **tr_attrs)
    tt.transforms = tt  # `from torchvision.transforms import transforms`
    tv = mod("torchvision", transforms=tt)
    tv.transforms.functional = tf  # `torchvision.transforms.functional.resize` (stylex_train.py:482)
    tv.utils = types.SimpleNamespace(save_image=lambda *a, **k: None, make_grid=lambda x, **k: x)
    tv.datasets = types.SimpleNamespace()
    mod("retry", api=None)
    mod("retry.api", retry_call=lambda f, **k: f())


_ST = None
_ST_NEW = None


def import_reference_new():
    """The reference's second architecture module ``stylex_train_new`` (cli.py:17-22 switch), imported on CPU with
    the same stubs."""
    global _ST_NEW
    if _ST_NEW is not None:
        return _ST_NEW
    import_reference()  # stubs, .cuda() patches
    real_avail = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    seed_state = torch.random.get_rng_state()
    try:
        torch.manual_seed(1234)
        sys.path.insert(0, REF_STYLEX)
        import stylex_train_new as stn  # noqa
    finally:
        torch.cuda.is_available = real_avail
        torch.random.set_rng_state(seed_state)
        if REF_STYLEX in sys.path:
            sys.path.remove(REF_STYLEX)
    _ST_NEW = stn
    return stn


def import_reference():
    """Return the reference's ``stylex_train`` module, imported once on CPU."""
    global _ST
    if _ST is not None:
        return _ST
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    real_avail = torch.cuda.is_available
    torch.cuda.is_available = lambda: True  # import-time assert, stylex_train.py:51
    seed_state = torch.random.get_rng_state()
    try:
        torch.manual_seed(1234)  # module-level lpips stand-in draws weights at import (:404)
        sys.path.insert(0, REF_STYLEX)
        import stylex_train as st  # noqa
    finally:
        torch.cuda.is_available = real_avail if not callable(real_avail) else (lambda: False)
        torch.random.set_rng_state(seed_state)
        if REF_STYLEX in sys.path:
            sys.path.remove(REF_STYLEX)
    _ST = st
    return st


def make_reference_trainer(st, base_dir, classifier, batches, **kw):
    """Build the reference ``Trainer`` wired to in-memory data."""
    st.ResNet = lambda *a, **k: classifier
    st.MobileNet = lambda *a, **k: classifier
    args = dict(name="gold", base_dir=base_dir, classifier_name="resnet", classifier_path="x",
                tensorboard_dir=None, evaluate_every=10 ** 9, save_every=10 ** 9)
    args.update(kw)
    tr = st.Trainer(**args)
    tr.loader = st.cycle(batches)
    tr.dataset = list(range(1000))
    tr.save = lambda *a, **k: None
    tr.evaluate = lambda *a, **k: None
    return tr


from standins import TinyClassifier, _load_by_path, _tv_models, seeded_mobilenet_state, seeded_resnet_state  # noqa: E402,F401


def import_reference_mobilenet():
    """The reference's own ``MobileNet`` wrapper class (stylex/mobilenet_classifier.py:28-73) with
    ``torch.hub.load`` replaced by the in-repo architecture (no network)."""
    import_reference()  # installs the torchvision stubs
    MobileNetV2 = _tv_models().MobileNetV2
    torch.hub.load = lambda *a, **k: MobileNetV2()
    return _load_by_path("ref_mobilenet_classifier", os.path.join(REF_STYLEX, "mobilenet_classifier.py"))


def import_reference_resnet():
    """The reference's own ``ResNet`` wrapper class (stylex/resnet_classifier.py:29-71) with ``torch.hub.load``
    replaced by the in-repo ResNet-18 architecture (no network).  ``resize`` / ``Normalize`` are the torchvision
    restatements of ``_install_stubs`` (bilinear, align_corners=False, no antialias on tensors in 0.11.1)."""
    import_reference()  # installs the torchvision stubs
    ResNet18 = _tv_models().ResNet18
    torch.hub.load = lambda *a, **k: ResNet18()
    return _load_by_path("ref_resnet_classifier", os.path.join(REF_STYLEX, "resnet_classifier.py"))
