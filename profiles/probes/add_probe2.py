"""Which tensors do the in-place adds issued from inside D's backward touch?  (shapes + parent chain)"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train(); torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name not in ("aten::add_", "aten::sum", "aten::mul", "aten::add") or (e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::")):
        continue
    fr = [f for f in (e.stack or []) if "stylex" in f or "ops.py" in f or "networks.py" in f]
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 3:
        chain.append(p.name[-50:]); p = p.cpu_parent
    c[(e.name, str(e.input_shapes)[:70], (fr[0][-40:] if fr else "?"), " < ".join(chain))] += 1
for k, n in c.most_common(60):
    print(n, *k, sep=" | ")
