"""Micro-benchmark of the generator's modulated convs as the train step launches them (B=64, bf16): forward with
modulation (in_scale), demodulation (out_scale), transposed noise and LeakyReLU vs the same conv with parts of that
epilogue removed, and the modulated data gradient.  python tools/bench_modconv.py [--batch 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402

LAYERS = [("G4.conv2", 128, 128, 64), ("G5.conv1", 128, 64, 128), ("G5.conv2", 64, 64, 128), ("G6.conv1", 64, 32, 256),
          ("G6.conv2", 32, 32, 256)]


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
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    P, dev, B = hb.BF16_ACT, "cuda:0", a.batch
    print("%-9s %4s %4s %4s | %8s %8s %8s %8s %8s | %8s %8s" % ("layer", "C", "N", "res", "plain", "+scales", "+noise", "full", "full_nat", "dgrad", "dgrad_mod"))
    for name, c, n, res in LAYERS:
        x = torch.randn(B, c, res, res, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, n, res, res, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(n, c, 3, 3, device=dev) * 0.05)
        s1, d = torch.rand(B, c, device=dev) + 0.5, torch.rand(B, n, device=dev) + 0.5
        noise = torch.rand(B, 256, 256, device=dev)
        nw, nb = torch.randn(n, device=dev), torch.randn(n, device=dev)
        t = [timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P)),
             timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P, in_scale=s1, out_scale=d)),
             timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P, noise=noise, noise_w=nw, noise_b=nb, lrelu=True)),
             timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P, in_scale=s1, out_scale=d, noise=noise, noise_w=nw, noise_b=nb, lrelu=True)),
             timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P, in_scale=s1, out_scale=d, noise=noise, noise_w=nw, noise_b=nb, lrelu=True,
                                          noise_natural=True)),
             timeit(lambda: hb.conv2d_bwd_data(dy, w, tuple(x.shape), 1, 1, P)),
             timeit(lambda: hb.conv2d_bwd_data(dy, w, tuple(x.shape), 1, 1, P, in_scale=d, out_scale=s1))]
        print("%-9s %4d %4d %4d | %8.3f %8.3f %8.3f %8.3f %8.3f | %8.3f %8.3f" % ((name, c, n, res) + tuple(t)))


if __name__ == "__main__":
    main()
