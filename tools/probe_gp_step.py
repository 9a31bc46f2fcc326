"""GPU box: what a gradient-penalty call adds.  Kernel time by name for a plain train() call and a penalty call of the bench
workload (torch profiler, device side), top differences.  python tools/probe_gp_step.py"""
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
for i in range(8):
    tr.train()
torch.cuda.synchronize()


def prof_call(step):
    tr.steps = step
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        tr.train()
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            rows[e.name[:90]][0] += 1
            rows[e.name[:90]][1] += e.device_time
    return rows


plain, gp = prof_call(9), prof_call(12)
tp, tg = sum(v[1] for v in plain.values()), sum(v[1] for v in gp.values())
print("kernel time: plain call %.2f ms (%d launches), penalty call %.2f ms (%d launches)" % (
    tp / 1e3, sum(v[0] for v in plain.values()), tg / 1e3, sum(v[0] for v in gp.values())))
diff = sorted(((gp[k][1] - plain.get(k, [0, 0.0])[1], k) for k in gp), reverse=True)
for d, k in diff[:28]:
    print("  %+8.1f us  %4d -> %4d  %s" % (d, plain.get(k, [0, 0])[0], gp[k][0], k))
