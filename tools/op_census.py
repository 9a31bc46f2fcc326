"""Which ATen ops does one steady-state train step launch (count, shapes)?  GPU box: python tools/op_census.py"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py"]
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16")
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
for i in range(6):
    tr.train()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::"):
        cnt[(e.name, str(e.input_shapes)[:110])] += 1
    elif e.name.startswith("aten::") and e.cpu_parent is None:
        cnt[(e.name, str(e.input_shapes)[:110])] += 1
tot = collections.Counter()
for (n, s), c in cnt.items():
    tot[n] += c
print("top-level aten ops per step:", sum(cnt.values()))
for n, c in tot.most_common(30):
    print("%6d %s" % (c, n))
print()
for (n, s), c in cnt.most_common(70):
    print("%5d %-28s %s" % (c, n, s))
