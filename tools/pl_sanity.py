"""Path-length-regularisation steps at full size in the bf16 speed mode (the reference only enables them after
step 5000, so the bench never reaches them): forces pl_after=4, pl_every=4 and reports step time, pl_mean and
losses.  python tools/pl_sanity.py"""
import argparse
import math
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
tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
tr.pl_after, tr.pl_every = 4, 4
for i in range(18):
    torch.cuda.synchronize()
    t = time.perf_counter()
    tr.train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) * 1e3
    pl = (tr.steps - 1) > tr.pl_after and (tr.steps - 1) % tr.pl_every == 0
    vals = (tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss)
    assert all(math.isfinite(v) for v in vals), vals
    print("step %2d %s %6.1f ms  pl_mean=%s  D=%.3g G=%.3g" % (i, "PL" if pl else "  ", dt, tr.pl_mean, vals[0], vals[1]))
print("peak device memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
