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
import torch
import torch.nn.functional as F
from torch import nn

# (cin, cout, kernel, stride, pad, maxpool_before, torchvision-features index)
_ALEX = [(3, 64, 11, 4, 2, False, 0), (64, 192, 5, 1, 2, True, 3), (192, 384, 3, 1, 1, True, 6),
         (384, 256, 3, 1, 1, False, 8), (256, 256, 3, 1, 1, False, 10)]


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
            x = F.relu(F.conv2d(x, self.cw[i], self.cb[i], stride=s, padding=p))
            out.append(x)
        return out

    def forward(self, in0, in1):
        f0 = self._taps((in0 - self.shift) / self.scale)
        f1 = self._taps((in1 - self.shift) / self.scale)
        total = 0
        for i in range(len(_ALEX)):
            n0 = f0[i] / (f0[i].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            n1 = f1[i] / (f1[i].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            total = total + F.conv2d((n0 - n1) ** 2, self.lin[i]).mean(dim=(2, 3), keepdim=True)
        return total
