"""GPU box: which kernels run (name, us) for the two frozen stems — forward / input gradient, plain F.conv2d vs
frozen_resnet.first_conv — in immediate mode as bench.py sets it.  python tools/probe_first_conv.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

import hip_backend as hb
from frozen_resnet import first_conv

hb.load_library()
torch.backends.cudnn.benchmark = False
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)


def kernels(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            r = rows.setdefault(e.name[:90], [0, 0.0])
            r[0] += 1
            r[1] += e.device_time
    for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("      %7.1f us x %4.1f  %s" % (t / c, c / reps, n))


for name, (N, K, S, P, H) in {"LPIPS-AlexNet conv1 11x11/4": (64, 11, 4, 2, 256), "ResNet conv1 7x7/2": (64, 7, 2, 3, 224)}.items():
    w = torch.randn(N, 3, K, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    x = torch.randn(32, 3, H, H, device=dev, generator=g)
    print("==", name)
    print("   forward, no grad (plain F.conv2d):")
    kernels(lambda: F.conv2d(x, w, b, S, P))
    for mode in ("0", "2"):
        os.environ["STYLEX_IMAGE_GRAD"] = mode
        xr = x.clone().requires_grad_()
        print("   forward + input gradient, STYLEX_IMAGE_GRAD=%s:" % mode)

        def fb():
            y = first_conv(xr, w, b, S, P)
            y.sum().backward()
            xr.grad = None

        kernels(fb)
