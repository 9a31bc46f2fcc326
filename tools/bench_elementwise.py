"""Achieved HBM bandwidth of the memory-bound kernels (blur, upsample, activation-backward reductions) on the
256 px StylEx activation shapes.  Usage (GPU box): python tools/bench_elementwise.py [--batch 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402


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
    dev = "cuda:0"
    dt = torch.bfloat16
    print("%-22s %-22s %9s %9s" % ("kernel", "shape", "ms", "GB/s"))
    for (c, r) in ((64, 256), (32, 256), (128, 128), (64, 128), (256, 64), (512, 32), (512, 16), (512, 8)):
        b = a.batch
        x = torch.randn(b, c, r, r, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        y = torch.randn_like(x)
        x2 = torch.randn(b, 4 * c, r // 2, r // 2, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        nb = x.numel() * 2.0
        s = torch.rand(b, c, device=dev) + 0.5
        plane = torch.rand(b, r, r, device=dev)
        nw, nbias = torch.randn(c, device=dev), torch.randn(c, device=dev)
        cases = [
            ("blur3x3_fwd", lambda: hb.blur3x3_fwd(x), 2 * nb),
            ("blur3x3_bwd", lambda: hb.blur3x3_bwd(x), 2 * nb),
            ("blur3x3_s2d_fwd", lambda: hb.blur3x3_s2d_fwd(x), 2 * nb),
            ("blur3x3_s2d_bwd", lambda: hb.blur3x3_s2d_bwd(x2), 2 * nb),
            ("act_bwd_reduce", lambda: hb.act_bwd_reduce(x, y, True, 1.0, want_dx=True), 3 * nb),
            ("act_bwd_reduce(nodx)", lambda: hb.act_bwd_reduce(x, None, False, 1.0, want_dx=False), 1 * nb),
            ("modconv_bwd_prep", lambda: hb.modconv_bwd_prep(x, y, plane, nw, nbias, True), 3 * nb),
            ("scale_reduce", lambda: hb.scale_reduce(x, y, s, want_gx=True), 3 * nb),
            ("bias_act_bwd", lambda: hb.bias_act_bwd(x, y), 3 * nb),
        ]
        w3 = torch.randn(3, c, 1, 1, device=dev)
        g4 = torch.randn(b, 4, r, r, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        rgbb = b * r * r * 8.0
        cases += [("torgb_fwd", lambda: hb.torgb_fwd(x, s, w3), nb + rgbb),
                  ("torgb_bwd", lambda: hb.torgb_bwd(x, g4, s, w3), 2 * nb + rgbb)]
        if r <= 128:
            cases += [("upsample2x_fwd", lambda: hb.upsample2x_fwd(x), 5 * nb),
                      ("upsample2x_bwd", lambda: hb.upsample2x_bwd(x), 1.25 * nb)]
        for name, fn, byts in cases:
            try:
                t = timeit(fn)
            except Exception as e:  # noqa: BLE001
                print("%-22s %-22s failed: %s" % (name, (b, c, r, r), str(e)[:60]))
                continue
            print("%-22s %-22s %9.3f %9.0f" % (name, (b, c, r, r), t, byts / t / 1e6))


if __name__ == "__main__":
    main()
