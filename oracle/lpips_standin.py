"""TEST INFRASTRUCTURE (oracle side).

Restatement of the published LPIPS-AlexNet metric (Zhang et al. 2018; package
``lpips==0.1.4`` pinned by the reference's environment.yml:212, call sites
stylex/stylex_train.py:404,415).  The pretrained AlexNet / linear-head weights
are fetched from the network by the real package and are NOT available offline,
so this stand-in draws seeded random weights of the published shapes:
**parity unpinned** for the learned weights; the arithmetic graph is the
published one:

  x -> (x - shift) / scale
    -> AlexNet feature taps after each of the 5 ReLUs
    -> per-tap channel unit-normalisation  f / (||f||_2 + 1e-10)
    -> squared difference -> 1x1 conv (non-negative weights, no bias)
    -> spatial mean -> sum over taps          => [N,1,1,1]
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_ALEX_CFG = [  # (cin, cout, k, stride, pad, maxpool_before)
    (3, 64, 11, 4, 2, False),
    (64, 192, 5, 1, 2, True),
    (192, 384, 3, 1, 1, True),
    (384, 256, 3, 1, 1, False),
    (256, 256, 3, 1, 1, False),
]


class LPIPSStandIn(nn.Module):
    def __init__(self, net="alex", seed=4242, **_):
        super().__init__()
        assert net == "alex"
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer("scale", torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))
        self.cw = nn.ParameterList()
        self.cb = nn.ParameterList()
        self.lin = nn.ParameterList()
        for (ci, co, k, s, p, mp) in _ALEX_CFG:
            fan_in = ci * k * k
            self.cw.append(nn.Parameter(torch.randn(co, ci, k, k, generator=g) * (2.0 / fan_in) ** 0.5,
                                        requires_grad=False))
            self.cb.append(nn.Parameter(torch.zeros(co), requires_grad=False))
            self.lin.append(nn.Parameter(torch.rand(1, co, 1, 1, generator=g) / co, requires_grad=False))
        self.eval()

    def features(self, x):
        taps = []
        for i, (ci, co, k, s, p, mp) in enumerate(_ALEX_CFG):
            if mp:
                x = F.max_pool2d(x, 3, 2)
            x = F.relu(F.conv2d(x, self.cw[i], self.cb[i], stride=s, padding=p))
            taps.append(x)
        return taps

    def forward(self, a, b):
        fa = self.features((a - self.shift) / self.scale)
        fb = self.features((b - self.shift) / self.scale)
        total = 0
        for i, (xa, xb) in enumerate(zip(fa, fb)):
            na = xa / (xa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            nb = xb / (xb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            d = F.conv2d((na - nb) ** 2, self.lin[i])
            total = total + d.mean(dim=(2, 3), keepdim=True)
        return total
