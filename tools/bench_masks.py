"""Cost / gain of the activation bit masks per launch (GPU box): python tools/bench_masks.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
P = hb.BF16_ACT
dev = "cuda:0"


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


mk = lambda *sh: torch.randn(*sh, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
print("%-34s %9s %9s" % ("launch (B=64)", "tensor ms", "mask ms"))
for C, N, S in ((64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (8, 64, 256)):
    B = 64
    x = mk(B, C, S, S)
    w = torch.randn(N, C, 3, 3, device=dev) / (9 * C) ** 0.5
    bias = torch.randn(N, device=dev)
    t0 = timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True))
    t1 = timeit(lambda: hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True, want_mask=True))
    print("%-34s %9.3f %9.3f" % ("fwd %d->%d @%d (plain / +mask out)" % (C, N, S), t0, t1))
    y, m = hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True, want_mask=True)
    if C >= 64:
        w2 = torch.randn(N, N, 3, 3, device=dev) / (9 * N) ** 0.5
        dy = mk(B, N, S, S)
        tp = timeit(lambda: hb.conv2d_bwd_data(dy, w2, (B, N, S, S), 1, 1, P))
        t0 = timeit(lambda: hb.conv2d_bwd_data(dy, w2, (B, N, S, S), 1, 1, P, gate=y))
        t1 = timeit(lambda: hb.conv2d_bwd_data(dy, w2, (B, N, S, S), 1, 1, P, gate=y, gate_mask=m))
        print("%-34s %9.3f %9.3f   (ungated %.3f)" % ("dgrad %d->%d @%d gate" % (N, N, S), t0, t1, tp))
    dy2 = hb.blur3x3_s2d_fwd(mk(B, N, S, S))
    tp = timeit(lambda: hb.blur3x3_s2d_bwd(dy2))
    t0 = timeit(lambda: hb.blur3x3_s2d_bwd(dy2, gate=y))
    t1 = timeit(lambda: hb.blur3x3_s2d_bwd(dy2, gate_mask=m))
    print("%-34s %9.3f %9.3f   (ungated %.3f)" % ("blur s2d adjoint %d @%d gate" % (N, S), t0, t1, tp))
