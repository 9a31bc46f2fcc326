"""Which tensors do the aten::add_ calls of the discriminator backward touch?  (shapes + the autograd node that issued them)"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]; sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
import bench
a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16")
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb, ops
hb.load_library(); ops.set_precision("bf16")
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
for i in range(6): tr.train()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    tr.train(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add_", "aten::add", "aten::mul", "aten::sum", "aten::copy_", "aten::to", "aten::contiguous", "aten::cat", "aten::fill_", "aten::zero_", "aten::clone", "aten::empty_like", "aten::zeros_like"):
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        if par.startswith("aten::"): continue
        cnt[(e.name, par[:60], str(e.input_shapes)[:70])] += 1
for k, v in cnt.most_common(70):
    print("%4d  %-14s %-62s %s" % (v, k[0], k[1], k[2]))
