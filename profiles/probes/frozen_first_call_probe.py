"""Is a frozen MIOpen network bit-reproducible call to call inside one process?  Same weights, same input, forward and
input gradient, repeated; cudnn.deterministic pinned as in tools/determinism_check.py.
usage (GPU box): python tools/probes/frozen_first_call_probe.py"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402

argv, sys.argv = sys.argv, ["bench.py"]
import bench  # noqa: E402

sys.argv = argv
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402
import stylex_train as st  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
torch.backends.cudnn.deterministic = True
dev = torch.device("cuda:0")
a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb_fp", precision="bf16")
g = torch.Generator().manual_seed(3)
for t in range(2):
    bench.seed_all(42)
    tr = bench.build_trainer(a, dev, 0, 1)
    for bsz in (32, 64):
        x0 = torch.rand(bsz, 3, 256, 256, generator=torch.Generator().manual_seed(5)).to(dev)
        outs, grads, lp = [], [], []
        for k in range(4):
            x = x0.clone().requires_grad_()
            y = tr._classify(x)
            (gx,) = torch.autograd.grad(y.square().sum(), x)
            outs.append(y.detach().double().sum().item())
            grads.append(gx.double().abs().sum().item())
            x2 = x0.clone().requires_grad_()
            p = st.perceptual_loss(x0.flip(0), x2, tr.lpips_fn)
            (g2,) = torch.autograd.grad(p.sum(), x2)
            lp.append((p.detach().double().sum().item(), g2.double().abs().sum().item()))
        print("trainer", t, "B", bsz, "classifier fwd", ["%.10f" % v for v in outs])
        print("trainer", t, "B", bsz, "classifier dgrad", ["%.8f" % v for v in grads])
        print("trainer", t, "B", bsz, "lpips", ["%.10f / %.8f" % v for v in lp])
    del tr
    torch.cuda.empty_cache()
