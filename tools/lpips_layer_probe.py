"""LPIPS-AlexNet layer by layer at the bench's batch (2 x 32 images, 256 px): forward + data gradient on stock PyTorch
(MIOpen fp32, what lpips_alex.py runs today) against ops.conv2d on the HIP bf16 kernels.
GPU box: python tools/lpips_layer_probe.py [--batch 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

ALEX = [(3, 64, 11, 4, 2, 256), (64, 192, 5, 1, 2, 31), (192, 384, 3, 1, 1, 15), (384, 256, 3, 1, 1, 15), (256, 256, 3, 1, 1, 15)]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    hb.load_library()
    ops.set_precision("bf16")
    torch.backends.cudnn.benchmark = True
    dev = "cuda:0"
    print("%-22s %5s | %9s %9s | %9s %9s | GF" % ("layer", "B", "torch fwd", "torch bwd", "hip fwd", "hip bwd"))
    tot = [0.0] * 4
    for (ci, co, k, s, p, res) in ALEX:
        for b in (a.batch, 2 * a.batch):
            w = torch.randn(co, ci, k, k, device=dev) * (2.0 / (ci * k * k)) ** 0.5
            bias = torch.zeros(co, device=dev)
            x = torch.randn(b, ci, res, res, device=dev, requires_grad=True)
            y = F.relu(F.conv2d(x, w, bias, stride=s, padding=p))
            gy = torch.randn_like(y)
            t_f = timeit(lambda: F.relu(F.conv2d(x, w, bias, stride=s, padding=p)))
            t_b = timeit(lambda: torch.autograd.grad(F.relu(F.conv2d(x, w, bias, stride=s, padding=p)), x, gy)) - t_f
            h_f = h_b = float("nan")
            try:
                cp = 8 if ci == 3 else ci
                xh = torch.randn(b, cp, res, res, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
                wh = torch.zeros(co, cp, k, k, device=dev)
                wh[:, :ci] = w
                prev = ops.set_fast(True)
                yh = ops.conv2d(xh, wh, bias, stride=s, padding=p, lrelu="relu")
                gyh = torch.randn_like(yh)
                h_f = timeit(lambda: ops.conv2d(xh, wh, bias, stride=s, padding=p, lrelu="relu"))
                h_b = timeit(lambda: torch.autograd.grad(ops.conv2d(xh, wh, bias, stride=s, padding=p, lrelu="relu"), xh, gyh)) - h_f
                ops.set_fast(prev)
            except Exception as e:  # noqa: BLE001
                print("   hip path failed:", repr(e)[:150])
            gf = 2.0 * b * y.shape[2] * y.shape[3] * co * ci * k * k / 1e9
            print("%2d->%3d k%-2d s%d @%-3d      %5d | %9.3f %9.3f | %9.3f %9.3f | %.1f" % (ci, co, k, s, res, b, t_f, t_b, h_f, h_b, gf))
            if b == a.batch:
                for i, v in enumerate((t_f, t_b, h_f, h_b)):
                    tot[i] += v
    print("sum at B=%d (one of the two images): torch fwd %.3f bwd %.3f | hip fwd %.3f bwd %.3f ms" % (a.batch, *tot))


if __name__ == "__main__":
    main()
