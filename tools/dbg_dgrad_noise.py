"""How noisy is the bf16 mode's input gradient of D itself (the adversarial signal)?  Compared with fp32 mode."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402
import networks  # noqa: E402
import ops  # noqa: E402

hb.load_library()
DEV = "cuda:0"
for size in (64, 256):
    torch.manual_seed(0)
    D = networks.DiscriminatorE(size, network_capacity=16).to(DEV)
    x = torch.rand(4, 3, size, size, device=DEV)
    res = {}
    for prec in ("fp32", "bf16"):
        ops.set_precision(prec)
        for fast in (True,):
            ops.set_fast(fast)
            xx = x.clone().requires_grad_()
            D(xx).sum().backward()
            res[prec] = xx.grad.double()
    ops.set_fast(False)
    print(size, "D input-gradient: bf16 vs fp32 relative L2 error %.3e" % float((res["bf16"] - res["fp32"]).norm() / res["fp32"].norm()))
