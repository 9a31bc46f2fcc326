"""TEST INFRASTRUCTURE — seeded stand-ins for the frozen classifier weights (the real ``torch.hub`` weights and
``trained_classifiers/*.pt`` are unavailable offline).  Pure torch, no reference import and no monkey-patch: this is
the only oracle-side module besides ``stylex_oracle`` / ``lpips_standin`` that the ``-m gpu`` tests, ``smoke()`` and
``bench.py``'s ``cpu_baseline`` import on the GPU box (``oracle/ref_shim.py`` — which patches ``torch.Tensor.cuda`` to
import the reference — never is)."""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


class TinyClassifier:
    """Seeded stand-in for the frozen classifier object used by Trainer
    (reference: ResNet/MobileNet wrappers with ``classify_images``).  Used for
    step-parity fixtures, where the real torch.hub weights are unavailable."""

    def __init__(self, seed=99, num_classes=2, image_size=32):
        g = torch.Generator().manual_seed(seed)
        self.w1 = torch.randn(8, 3, 3, 3, generator=g) * 0.3
        self.b1 = torch.randn(8, generator=g) * 0.1
        self.w2 = torch.randn(num_classes, 8, generator=g) * 0.5
        self.b2 = torch.randn(num_classes, generator=g) * 0.1
        self.mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        self.std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        self.image_size = image_size

    def to(self, device):
        for n in ("w1", "b1", "w2", "b2", "mean", "std"):
            setattr(self, n, getattr(self, n).to(device))
        return self

    def classify_images(self, images):
        x = (images - self.mean) / self.std
        x = F.leaky_relu(F.conv2d(x, self.w1, self.b1, stride=2, padding=1), 0.2)
        x = x.mean(dim=(2, 3))
        return x @ self.w2.t() + self.b2


def _load_by_path(name, path):
    import importlib.util

    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _tv_models():
    """The product's in-repo torchvision-compatible architecture definitions, loaded by path (its directory is not
    put on sys.path: module names there collide with the reference's)."""
    return _load_by_path("stylex_amd_tv_models", os.path.join(os.path.dirname(_HERE),
                                                             "explaining-in-style-reproducibility-study_amd", "stylex",
                                                             "tv_models.py"))


def seeded_mobilenet_state(seed=77, output_size=2):
    """State dict of a MobileNetV2 (torchvision key layout, 2-logit head) with seeded random weights AND non-trivial
    BatchNorm statistics — the stand-in for ``trained_classifiers/<name>`` in the config-4 fixtures (the real
    torch.hub weights / checkpoints are unavailable offline).  Uses the in-repo architecture definition
    (tv_models.MobileNetV2): weights only; the forward that the golden records is the reference wrapper's."""
    MobileNetV2 = _tv_models().MobileNetV2
    state = torch.random.get_rng_state()
    try:
        torch.manual_seed(seed)
        model = MobileNetV2()
        model.classifier[1] = nn.Linear(1280, output_size)
        g = torch.Generator().manual_seed(seed + 1)
        for m in model.modules():
            if isinstance(m, nn.Conv2d):  # He init: default-initialised, the 53-layer net maps every input to the same logits
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                with torch.no_grad():
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
                with torch.no_grad():
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        with torch.no_grad():
            model.classifier[1].weight.copy_(torch.randn(output_size, 1280, generator=g) * 0.05)
    finally:
        torch.random.set_rng_state(state)
    return model.state_dict()


def seeded_resnet_state(seed=55, output_size=2):
    """State dict of a ResNet-18 (torchvision key layout, 2-logit head, as ``load_resnet_classifier`` builds it,
    stylex/resnet_classifier.py:19-22) with seeded He-initialised weights and non-trivial BatchNorm statistics /
    affine parameters — the stand-in for ``trained_classifiers/<name>`` in the A16 fixture."""
    ResNet18 = _tv_models().ResNet18
    state = torch.random.get_rng_state()
    try:
        torch.manual_seed(seed)
        model = ResNet18()
        model.fc = nn.Linear(512, output_size)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if isinstance(m, nn.BatchNorm2d):
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                    m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            model.fc.weight.copy_(torch.randn(output_size, 512, generator=g) * 0.05)
            model.fc.bias.copy_(torch.randn(output_size, generator=g) * 0.1)
    finally:
        torch.random.set_rng_state(state)
    return model.state_dict()
