"""Per-layer micro-benchmark of the conv kernels on the 256 px / B=32 StylEx shapes.
Usage (GPU box): python tools/bench_conv.py [--precision bf16|fp32] [--batch 32] [--size 256]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402
from networks import discriminator_filters, generator_filters  # noqa: E402


def layers(size, cap=16, fmax=512):
    out = []
    gf = generator_filters(size, cap, fmax)
    for i in range(len(gf) - 1):
        res = 4 * 2 ** i
        out.append(("G%d.conv1" % i, gf[i], gf[i + 1], res, 3, 1, 1))
        out.append(("G%d.conv2" % i, gf[i + 1], gf[i + 1], res, 3, 1, 1))
        out.append(("G%d.rgb" % i, gf[i + 1], 3, res, 1, 1, 0))
    df = discriminator_filters(size, cap, fmax)
    for i in range(len(df) - 1):
        res = size // 2 ** i
        last = i == len(df) - 2
        out.append(("D%d.res" % i, df[i], df[i + 1], res, 1, 1 if last else 2, 0))
        out.append(("D%d.conv1" % i, df[i], df[i + 1], res, 3, 1, 1))
        out.append(("D%d.conv2" % i, df[i + 1], df[i + 1], res, 3, 1, 1))
        if not last:
            out.append(("D%d.down" % i, df[i + 1], df[i + 1], res, 3, 2, 1))
    return out


def timeit(fn, iters):
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
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    prec = {"bf16": hb.BF16_ACT, "bf16_f32act": hb.BF16, "fp32": hb.F32}[a.precision]
    adt = hb.act_dtype(prec)
    dev = "cuda:0"
    print("%-10s %5s %5s %4s k s | %9s %9s %9s | TF/s fwd dgrad wgrad | GB/s fwd" % ("layer", "C", "N", "res", "fwd ms", "dgrad ms", "wgrad ms"))
    tot = [0.0, 0.0, 0.0]
    totf = 0.0
    for (name, c, n, res, k, s, p) in layers(a.size):
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        cp = c
        if c == 3:  # the product pads RGB to one 16-byte slot (ops._pad_rgb)
            cp = 8 if adt == torch.bfloat16 else 4
        npad = 4 if n == 3 else n
        w = torch.randn(npad, cp, k, k, device=dev) * 0.05
        if k == 3 and s == 2 and prec == hb.BF16_ACT and res // 2 >= 16 and c % 64 == 0:
            # the product's down-sampling conv: space-to-depth input, 3x3/s1 over 4C channels, masked taps
            x = torch.randn(a.batch, 4 * c, res // 2, res // 2, device=dev).to(adt).contiguous(memory_format=torch.channels_last)
            wf2, wb2 = hb.pack_weight_s2d(w)
            ws = (n, 4 * c, 3, 3)
            fwd = lambda: hb.conv2d_fwd(x, None, 1, 1, prec, packed=wf2, w_shape=ws, s2d_c=c)
            y = fwd()
            dy = torch.randn_like(y)
            bwd = lambda: hb.conv2d_bwd_data(dy, None, tuple(x.shape), 1, 1, prec, packed=wb2, w_shape=ws, s2d_c=c)
            wgr = lambda: hb.conv2d_bwd_weight_s2d(x, dy, (n, c, 3, 3), prec)  # the product's call: folded layout in one launch pair
        else:
            x = torch.randn(a.batch, cp, res, res, device=dev).to(adt).contiguous(memory_format=torch.channels_last)
            fwd = lambda: hb.conv2d_fwd(x, w, s, p, prec)
            y = fwd()
            dy = torch.randn_like(y)
            bwd = lambda: hb.conv2d_bwd_data(dy, w, tuple(x.shape), s, p, prec)
            wgr = lambda: hb.conv2d_bwd_weight(x, dy, tuple(w.shape), s, p, prec)
        t_f, t_d, t_w = timeit(fwd, a.iters), timeit(bwd, a.iters), timeit(wgr, a.iters)
        fl = 2.0 * a.batch * n * (y.shape[2] * y.shape[3]) * c * k * k
        byts = float(x.element_size()) * (x.numel() + y.numel()) + 4.0 * w.numel()
        print("%-10s %5d %5d %4d %d %d | %9.3f %9.3f %9.3f | %6.1f %6.1f %6.1f | %7.0f" % (
            name, c, n, res, k, s, t_f, t_d, t_w, fl / t_f / 1e9, fl / t_d / 1e9, fl / t_w / 1e9, byts / t_f / 1e6))
        tot[0] += t_f; tot[1] += t_d; tot[2] += t_w; totf += fl
    print("TOTAL ms fwd %.2f dgrad %.2f wgrad %.2f ; GF %.1f ; TF/s %.1f %.1f %.1f" % (
        tot[0], tot[1], tot[2], totf / 1e9, totf / tot[0] / 1e9, totf / tot[1] / 1e9, totf / tot[2] / 1e9))


if __name__ == "__main__":
    main()
