"""GPU box: kernels (name, us, calls) of one frozen-classifier forward + input gradient at the bench shape (B = 32, 256 px),
bf16 speed mode (hybrid backward) and fp32 mode.  python tools/probe_classifier.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import torch
from torch.profiler import ProfilerActivity, profile

import hip_backend as hb
import ops
from resnet_classifier import ResNet

hb.load_library()
torch.backends.cudnn.benchmark = False
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
fake = torch.rand(32, 3, 256, 256, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
clf = ResNet(None, 0, output_size=2, image_size=256)


def step():
    x = fake.clone().requires_grad_()
    clf.classify_images(x).sum().backward()


for prec in ("bf16", "fp32"):
    ops.set_precision(prec)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    reps = 5
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            r = rows.setdefault(e.name[:100], [0, 0.0])
            r[0] += 1
            r[1] += e.device_time
    tot = sum(t for _, t in rows.values()) / reps
    print("== %s: %.1f us of kernels per forward + backward, %d launches" % (prec, tot, sum(c for c, _ in rows.values()) / reps))
    for n, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:32]:
        print("   %8.1f us/call x %5.1f = %8.1f us  %s" % (t / c, c / reps, t / reps, n))
