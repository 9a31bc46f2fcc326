"""Weight gradient of the 1x1 residual convs of the DiscriminatorBlocks (contiguous even-pixel input): LDS-DMA variant of the
general kernel (conv_wgrad_tr_dma_kernel) against the register-staged one (STYLEX_WGRAD_TR_DMA=0), checked against fp64.
GPU box: python tools/bench_wgrad_1x1.py [--batch 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402

SHAPES = [(64, 128, 64), (128, 256, 32), (256, 512, 16), (512, 512, 8), (512, 512, 4), (512, 512, 2), (512, 512, 1)]


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
print("%-22s %5s | %9s %9s | speed-up | rel err new / old" % ("layer", "B", "new ms", "old ms"))
for (c, n, res) in SHAPES:
    for b in (a.batch // 2, a.batch, 2 * a.batch):
        x = torch.randn(b, c, res, res, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        dy = torch.randn(b, n, res, res, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        ref = torch.einsum("bnyx,bcyx->nc", dy.double(), x.double())
        t, err = {}, {}
        for mode in ("1", "0", "1", "0"):
            os.environ["STYLEX_WGRAD_TR_DMA"] = mode
            fn = lambda: hb.conv2d_bwd_weight(x, dy, (n, c, 1, 1), 1, 0, hb.BF16_ACT)
            err[mode] = float((fn().double().reshape(n, c) - ref).abs().max() / ref.abs().max())
            t[mode] = min(t.get(mode, 1e9), timeit(fn))
        assert err["1"] < 2e-5, err
        print("%4d->%4d 1x1 @%-3d      %5d | %9.3f %9.3f | %.2fx | %.1e / %.1e" % (c, n, res, b, t["1"], t["0"], t["0"] / t["1"], err["1"], err["0"]))
os.environ["STYLEX_WGRAD_TR_DMA"] = "1"
