"""GPU box: full Python stacks and input shapes of the aten::add_ calls of one train() call.  python tools/probe_add_sites.py"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py"]
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench

a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16", device_rng=1)
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb
import ops

hb.load_library()
ops.set_precision("bf16")
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
for i in range(6):
    tr.train()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    tr.train()
    torch.cuda.synchronize()
cnt = collections.Counter()
first = {}
for e in prof.events():
    if e.name != "aten::add_" or (e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::")):
        continue
    par, chain = e.cpu_parent, []
    while par is not None:
        chain.append(par.name[:60])
        par = par.cpu_parent
    key = (str(e.input_shapes)[:60], " <- ".join(chain[:4]), " | ".join((e.stack or [])[:3]))
    cnt[key] += 1
for k, c in cnt.most_common(12):
    print(c, k)
