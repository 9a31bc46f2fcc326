"""Frozen MobileNetV2 classifier wrapper (reference API: stylex/mobilenet_classifier.py:28-73)."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from tv_models import MobileNetV2

_MEAN, _STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def load_classifier(model_name, cuda_rank, output_size=2, seed=1234):
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = MobileNetV2()
    model.classifier[1] = nn.Linear(1280, output_size)
    torch.random.set_rng_state(state)
    if model_name is not None:
        # a named checkpoint must exist, like the reference (torch.load raises FileNotFoundError): silently training
        # against a random classifier because of a typo / wrong cwd would be worse than stopping
        path = os.path.join("trained_classifiers", str(model_name))
        if not os.path.isfile(path):
            raise FileNotFoundError("classifier checkpoint %r not found (cwd %s); pass classifier_path=None for the "
                                    "seeded random-weight classifier of the synthetic benchmarks" % (path, os.getcwd()))
        model.load_state_dict(torch.load(path, map_location="cpu"))
    dev = torch.device("cuda:%d" % cuda_rank) if torch.cuda.is_available() else torch.device("cpu")
    return model.to(dev)


class MobileNet:
    def __init__(self, model_name, cuda_rank, output_size=2, image_size=32, normalize=True):
        self.model = load_classifier(model_name, cuda_rank, output_size)
        self.mobilenet_dim = 224
        self.image_size = image_size
        self.normalize = normalize
        dev = next(self.model.parameters()).device
        self._mean = torch.tensor(_MEAN, device=dev).view(1, 3, 1, 1)
        self._std = torch.tensor(_STD, device=dev).view(1, 3, 1, 1)
        for p in self.model.parameters():
            p.requires_grad = False
        self.model.eval()

    def classify_images(self, images):
        x = F.interpolate(images, size=self.image_size)  # nearest; identity at native size (:62)
        if self.normalize:
            x = (x - self._mean) / self._std
        return self.model(x)
