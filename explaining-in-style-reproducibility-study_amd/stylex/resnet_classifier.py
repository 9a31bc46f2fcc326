"""Frozen ResNet-18 classifier wrapper (reference API: stylex/resnet_classifier.py:29-71).
Forward stays on stock PyTorch-ROCm; gradients still flow to the input images."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from tv_models import ResNet18

_MEAN, _STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _device(cuda_rank):
    return torch.device("cuda:%d" % cuda_rank) if torch.cuda.is_available() else torch.device("cpu")


def load_resnet_classifier(model_name, cuda_rank, output_size=2, seed=1234):
    """ResNet-18 with a 2-logit head.  ``trained_classifiers/<model_name>`` is loaded (reference :16-26) and must
    exist; ``model_name=None`` is the explicit opt-in for seeded random weights (synthetic benchmarks — the
    torch.hub / checkpoint files are unavailable offline)."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = ResNet18()
    model.fc = nn.Linear(512, output_size)
    torch.random.set_rng_state(state)
    if model_name is not None:
        # a named checkpoint must exist, like the reference (torch.load raises FileNotFoundError): silently training
        # against a random classifier because of a typo / wrong cwd would be worse than stopping
        path = os.path.join("trained_classifiers", str(model_name))
        if not os.path.isfile(path):
            raise FileNotFoundError("classifier checkpoint %r not found (cwd %s); pass classifier_path=None for the "
                                    "seeded random-weight classifier of the synthetic benchmarks" % (path, os.getcwd()))
        model.load_state_dict(torch.load(path, map_location="cpu"))
    return model.to(_device(cuda_rank))


class _ResizeNormalize(torch.autograd.Function):
    """resize to `size` (bilinear, align_corners=False) + (x - mean) / std as one kernel each way (csrc/frozen_ew.hip,
    stylex_resize_norm_fwd / _bwd; round 6) — ATen's upsample kernel alone took 140-160 us per call and direction at B = 32.
    The input is read through its strides: the generator's channels_last batch needs no dense copy first."""

    @staticmethod
    def forward(ctx, x, size, mean, std):
        import hip_backend as hb

        ctx.hw, ctx.std = tuple(x.shape[2:]), std
        return hb.resize_norm_fwd(x, size, mean, std)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        import hip_backend as hb

        return hb.resize_norm_bwd(gy, ctx.hw, ctx.std), None, None, None


class ResNet:
    accepts_any_layout = True  # classify_images reads its input through strides on the GPU (Trainer._classify skips the dense copy)

    def __init__(self, model_name, cuda_rank, output_size=2, image_size=32, normalize=True):
        self.model = load_resnet_classifier(model_name, cuda_rank, output_size)
        self.resnet_dim = 224
        self.image_size = image_size
        self.normalize = normalize
        dev = next(self.model.parameters()).device
        self._mean = torch.tensor(_MEAN, device=dev).view(1, 3, 1, 1)
        self._std = torch.tensor(_STD, device=dev).view(1, 3, 1, 1)
        for p in self.model.parameters():
            p.requires_grad = False
        self.model.eval()

    def classify_images(self, images):
        # torchvision 0.11 resize on tensors == bilinear, align_corners=False, no antialias (:61)
        if (images.is_cuda and images.dtype == torch.float32 and images.dim() == 4
                and os.environ.get("STYLEX_RESIZE_FUSE", "1") != "0"):
            mean, std = (self._mean.reshape(-1), self._std.reshape(-1)) if self.normalize else (None, None)
            x = _ResizeNormalize.apply(images, (self.resnet_dim, self.resnet_dim), mean, std)
            return self._net(x)(x)
        x = F.interpolate(images.contiguous(), size=[self.resnet_dim, self.resnet_dim], mode="bilinear", align_corners=False)
        if self.normalize:
            x = (x - self._mean) / self._std
        return self._net(x)(x)

    def _net(self, x):
        """The stock PyTorch fp32 module (north_star: the frozen classifier's forward stays on stock PyTorch): on the GPU its
        library convolutions with everything between them (eval BatchNorm, ReLU, residual add, max-pool) on the fused fp32
        kernels of csrc/frozen_ew.hip, and — bf16 speed mode, a pass whose input gradient is wanted — the data gradient on
        this library's bf16 kernels gated by the fp32 activations' signs (frozen_resnet.FusedTailResNet).
        STYLEX_FROZEN_FUSE=0 = the plain nn.Module.  (A bf16 FORWARD of the classifier — the opt-in of rounds 1-5 — was
        deleted in round 6: its flipped ReLU gates put 20 % of L2 error on the input gradient.)"""
        if not x.is_cuda or os.environ.get("STYLEX_FROZEN_FUSE", "1") == "0":
            return self.model
        if getattr(self, "_fused", None) is None:
            from frozen_resnet import FusedTailResNet

            self._fused = FusedTailResNet(self.model) if FusedTailResNet.supports(self.model) else self.model
        return self._fused
