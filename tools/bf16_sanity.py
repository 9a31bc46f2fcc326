"""Speed-mode sanity: N train() calls of the bench workload (256 px, B=32, GAE=2, bf16), losses every 10 steps and
peak device memory — no NaN / blow-up, finite losses.  python tools/bf16_sanity.py [steps] [bf16|fp32]"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py"] + sys.argv[1:]
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"  # "fp32": the parity mode on the same workload, for comparison
a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision=prec)
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision(prec)
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
bad = 0
for i in range(steps):
    tr.train()
    if i % 10 == 0 or i == steps - 1:
        vals = (tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss, tr.last_gp_loss)
        bad += sum(0 if (v is None or math.isfinite(v)) else 1 for v in vals)
        print(i, " ".join("%s=%.4g" % (k, v if v is not None else float("nan")) for k, v in zip("D G rec kl gp".split(), vals)))
print("peak device memory %.1f GB; non-finite loss readings: %d" % (torch.cuda.max_memory_allocated() / 2 ** 30, bad))
