"""Which tensors do the ~150 in-place adds of a train() call touch?  (torch profiler, record_shapes, one plain call.)
python tools/probes/add_shapes_probe.py [op=aten::add_]"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
want = sys.argv[1] if len(sys.argv) > 1 else "aten::add_"
sys.argv = ["bench.py"]
import torch
import bench
a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16")
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb, ops
hb.load_library(); ops.set_precision("bf16")
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
for _ in range(6):
    tr.train()
torch.cuda.synchronize()
assert tr.steps % 4 != 0
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name == want:
        cnt[str(e.input_shapes)] += 1
for k, v in cnt.most_common(60):
    print(v, k)
print("total", sum(cnt.values()))
