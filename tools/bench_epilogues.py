"""Cost of every fused epilogue / staging option relative to the bare kernel, on the shapes the train step launches
(bf16, B=64 unless stated).  A large gap between two columns of a row = an epilogue that is not hidden behind the
conv.  python tools/bench_epilogues.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402

P, dev = hb.BF16_ACT, "cuda:0"


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


def t(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def main():
    B = 64
    print("== discriminator-type 3x3 convs: forward plain | +bias+lrelu ; dgrad plain | +gate ; wgrad")
    for name, c, n, res in [("D0.conv2", 64, 64, 256), ("D1.conv1", 64, 128, 128), ("D1.conv2", 128, 128, 128),
                            ("D2.conv2", 256, 256, 64), ("D3.conv2", 512, 512, 32), ("D4.conv2", 512, 512, 16)]:
        x, dy = t(B, c, res, res), t(B, n, res, res)
        w = torch.nn.Parameter(torch.randn(n, c, 3, 3, device=dev) * 0.05)
        b = torch.randn(n, device=dev)
        r = [timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P)), timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P, bias=b, lrelu=True)),
             timeit(lambda: hb.conv2d_bwd_data(dy, w, tuple(x.shape), 1, 1, P)),
             timeit(lambda: hb.conv2d_bwd_data(dy, w, tuple(x.shape), 1, 1, P, gate=x)),
             timeit(lambda: hb.conv2d_bwd_weight(x, dy, tuple(w.shape), 1, 1, P))]
        print("%-9s %4d %4d %4d | fwd %.3f %.3f | dgrad %.3f %.3f | wgrad %.3f" % ((name, c, n, res) + tuple(r)))
    print("== modulated generator convs: wgrad plain | with x_scale + dy_scale ; backward prep kernels")
    for name, c, n, res in [("G4.conv2", 128, 128, 64), ("G5.conv2", 64, 64, 128), ("G6.conv1", 64, 32, 256), ("G6.conv2", 32, 32, 256)]:
        x, dy = t(B, c, res, res), t(B, n, res, res)
        w = torch.nn.Parameter(torch.randn(n, c, 3, 3, device=dev) * 0.05)
        s1, d = torch.rand(B, c, device=dev) + 0.5, torch.rand(B, n, device=dev) + 0.5
        noise = torch.rand(B, 256, 256, device=dev)
        nw, nb = torch.randn(n, device=dev), torch.randn(n, device=dev)
        r = [timeit(lambda: hb.conv2d_bwd_weight(x, dy, tuple(w.shape), 1, 1, P)),
             timeit(lambda: hb.conv2d_bwd_weight(x, dy, tuple(w.shape), 1, 1, P, x_scale=s1, dy_scale=d)),
             timeit(lambda: hb.modconv_bwd_prep(dy, dy, noise, nw, nb, True)),
             timeit(lambda: hb.modconv_bwd_prep(dy, dy, None, None, None, True)),
             timeit(lambda: hb.scale_reduce(x, x, s1))]
        gb = 2.0 * dy.numel() * 2 / 1e9
        print("%-9s %4d %4d %4d | wgrad %.3f %.3f | bwd_prep %.3f (no noise %.3f; %.0f GB/s) | scale_reduce %.3f"
              % ((name, c, n, res) + tuple(r[:4]) + (3 * dy.numel() * 2 / r[2] / 1e6, r[4])))
    print("== stride-2 tail of a DiscriminatorBlock: blur s2d fwd | s2d conv plain | +bias+residual ; s2d dgrad | blur adjoint plain | +gate")
    for name, c, res in [("D0.down", 64, 256), ("D1.down", 128, 128), ("D2.down", 256, 64)]:
        y2 = t(B, c, res, res)
        w = torch.nn.Parameter(torch.randn(c, c, 3, 3, device=dev) * 0.05)
        wf2, wb2 = hb.pack_weight_s2d(w)
        xb = hb.blur3x3_s2d_fwd(y2)
        ws = (c, 4 * c, 3, 3)
        res_t, g = t(B, c, res // 2, res // 2), t(B, c, res // 2, res // 2)
        b = torch.randn(c, device=dev)
        r = [timeit(lambda: hb.blur3x3_s2d_fwd(y2)),
             timeit(lambda: hb.conv2d_fwd(xb, None, 1, 1, P, packed=wf2, w_shape=ws, s2d_c=c)),
             timeit(lambda: hb.conv2d_fwd(xb, None, 1, 1, P, packed=wf2, w_shape=ws, s2d_c=c, bias=b, residual=res_t, res_scale=0.7)),
             timeit(lambda: hb.conv2d_bwd_data(g, None, tuple(xb.shape), 1, 1, P, packed=wb2, w_shape=ws, s2d_c=c)),
             timeit(lambda: hb.blur3x3_s2d_bwd(xb)), timeit(lambda: hb.blur3x3_s2d_bwd(xb, gate=y2)),
             timeit(lambda: hb.conv2d_bwd_weight(xb, g, ws, 1, 1, P, s2d_c=c))]
        print("%-9s %4d %4d | blur %.3f | conv %.3f %.3f | dgrad %.3f | blur^T %.3f %.3f | wgrad %.3f" % ((name, c, res) + tuple(r)))
    print("== reductions / elementwise: act_bwd_reduce(sum only) | (dx+sum) ; upsample fwd | bwd ; torgb fwd | bwd")
    for c, res in [(64, 256), (128, 128), (256, 64)]:
        x = t(B, c, res, res)
        r = [timeit(lambda: hb.act_bwd_reduce(x, None, False, 1.0, want_dx=False, want_sum=True)),
             timeit(lambda: hb.act_bwd_reduce(x, x, True, 1.0, want_dx=True, want_sum=True))]
        half = t(B, c, res // 2, res // 2)
        r += [timeit(lambda: hb.upsample2x_fwd(half)), timeit(lambda: hb.upsample2x_bwd(x))]
        print("C=%3d %3d px | reduce %.3f (%.0f GB/s) %.3f | up %.3f (%.0f GB/s) %.3f" % (
            c, res, r[0], x.numel() * 2 / r[0] / 1e6, r[1], r[2], 1.25 * x.numel() * 2 / r[2] / 1e6, r[3]))


if __name__ == "__main__":
    main()
