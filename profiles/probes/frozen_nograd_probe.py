"""Opt-in STYLEX_FROZEN_HIP=nograd: how far are the bf16-kernel logits of a no-gradient classifier pass from the fp32
library ones (seeded ResNet-18 of the benchmark, 256 px inputs)?  GPU box: python tools/probes/frozen_nograd_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import argparse
import torch
sys.argv = ["bench.py"]
import bench
import hip_backend as hb, ops
hb.load_library(); ops.set_precision("bf16")
a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16")
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
x = torch.rand(32, 3, 256, 256, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))
with torch.no_grad():
    os.environ["STYLEX_FROZEN_HIP"] = "0"
    ref = tr._classify(x).double()
    os.environ["STYLEX_FROZEN_HIP"] = "nograd"
    got = tr._classify(x).double()
os.environ.pop("STYLEX_FROZEN_HIP")
d = (got - ref).abs()
print("logits fp32 library: mean |l| %.4g, spread over the batch %.4g" % (ref.abs().mean(), ref.std(dim=0).mean()))
print("bf16 kernels vs fp32: max |diff| %.4g, mean %.4g, relative to mean |l| %.3g, relative to the batch spread %.3g" % (
    d.max(), d.mean(), d.mean() / ref.abs().mean(), d.mean() / ref.std(dim=0).mean()))
p0, p1 = torch.softmax(ref, 1), torch.softmax(got, 1)
print("class probabilities: max |diff| %.4g" % (p0 - p1).abs().max())
