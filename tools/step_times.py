"""Per-step wall time of Trainer.train() on the bench workload, plus the host-only enqueue time of a step
(how long the CPU needs to launch it): if enqueue ~= wall the step is launch-bound, not GPU-bound."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py"]
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16")
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
dev = torch.device("cuda:0")
tr = bench.build_trainer(a, dev, 0, 1)
for i in range(4):
    tr.train()
tr.steps = 0
for i in range(8):
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.train()
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    print(i, "GP" if i % 4 == 0 else "  ", "wall %.1f ms   host enqueue %.1f ms" % ((time.perf_counter() - t) * 1e3, t_enq * 1e3))
# free-running (no per-step sync): what bench.py measures
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(8):
    tr.train()
torch.cuda.synchronize()
print("free-running: %.1f ms/step" % ((time.perf_counter() - t) * 1e3 / 8))
