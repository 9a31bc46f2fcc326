"""Which source lines of the train step launch the small ATen kernels (count and device time per call site)?
GPU box: python tools/op_sites.py [--top 60]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--by-count", action="store_true", help="rank call sites by number of calls instead of device time")
ap.add_argument("--by-func", action="store_true", help="aggregate per innermost package function (or autograd node) instead of per call chain")
ap.add_argument("--warm", type=int, default=6, help="train() calls before the profiled one (8 -> the profiled call is a penalty step)")
args = ap.parse_args()
sys.argv = ["bench.py"]
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16", device_rng=1)  # bench.py's defaults
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
for i in range(args.warm):
    tr.train()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    tr.train()
    torch.cuda.synchronize()

PKG = "explaining-in-style-reproducibility-study_amd"
OURS = set(f for f in os.listdir(os.path.join(ROOT, PKG, "stylex")) if f.endswith(".py")) | {"bench.py"}
site_t = collections.Counter()
site_n = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    dt = sum(k.duration for k in e.kernels) if hasattr(e, "kernels") else 0.0
    if dt <= 0:
        dt = getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0)
    site = "?"
    frames = [fr for fr in (e.stack or []) if fr.split("(")[0] in OURS]
    if frames:  # innermost package frames
        site = " <- ".join(fr[:44] for fr in frames[:1 if args.by_func else 3])
    else:
        par, node = e.cpu_parent, None
        while par is not None:  # an op issued by the autograd engine itself: name the node it runs under
            if "evaluate_function" in par.name or par.name.endswith("Backward0") or "Backward" in par.name:
                node = par.name
            par = par.cpu_parent
        if node:
            site = "[engine] " + node[-70:]
        elif e.stack:
            site = "[other] " + e.stack[0][-60:]
    site_t[(site, e.name)] += dt
    site_n[(site, e.name)] += 1
tot = sum(site_t.values())
print("device time of top-level aten ops: %.2f ms/step over %d calls" % (tot / 1e3, sum(site_n.values())))
SKIP = ("aten::conv2d", "aten::convolution_backward", "aten::batch_norm", "aten::native_batch_norm_backward")
order = sorted(site_t, key=lambda k: -site_n[k]) if args.by_count else [k for k, _ in site_t.most_common(args.top + 8)]
for (site, name) in order[:args.top + 8]:
    t = site_t[(site, name)]
    if name in SKIP or (args.by_func and t <= 0):
        continue
    print("%8.1f us %5d  %-26s %s" % (t, site_n[(site, name)], name, site))
