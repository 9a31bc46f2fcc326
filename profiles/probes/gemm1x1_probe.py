"""1x1 / stride-1 convs (the residual path of a DiscriminatorBlock after the even-pixel gather) are plain GEMMs
[M = B*H*W, K = C] x [K, N]: own generic kernel vs torch.mm / addmm (hipBLASLt) on the same bf16 tensors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch
import hip_backend as hb
P = hb.BF16_ACT
dev = "cuda:0"
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
print("B=%d   layer  C->N @res | own fwd  mm fwd | own dgrad  mm dgrad | own wgrad  mm wgrad (ms)" % B)
for (c, n, res) in [(64, 128, 64), (128, 256, 32), (256, 512, 16), (512, 512, 8), (512, 512, 4)]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, c, res, res, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(n, c, 1, 1, device=dev, generator=g) / c ** 0.5
    bias = torch.randn(n, device=dev, generator=g)
    y = hb.conv2d_fwd(x, w, 1, 0, P, bias=bias)
    dy = torch.randn_like(y)
    x2 = x.permute(0, 2, 3, 1).reshape(-1, c)
    dy2 = dy.permute(0, 2, 3, 1).reshape(-1, n)
    wb = w.view(n, c).to(torch.bfloat16)
    bb = bias.to(torch.bfloat16)
    f_own = lambda: hb.conv2d_fwd(x, w, 1, 0, P, bias=bias)
    f_mm = lambda: torch.addmm(bb, x2, wb.t())
    d_own = lambda: hb.conv2d_bwd_data(dy, w, tuple(x.shape), 1, 0, P)
    d_mm = lambda: torch.mm(dy2, wb)
    w_own = lambda: hb.conv2d_bwd_weight(x, dy, tuple(w.shape), 1, 0, P)
    w_mm = lambda: torch.mm(dy2.t(), x2)
    ym = f_mm().view(B, res, res, n).permute(0, 3, 1, 2)
    err = float((ym.float() - y.float()).abs().max()) / float(y.float().abs().max())
    print("%4d->%4d @%3d | %.3f  %.3f | %.3f  %.3f | %.3f  %.3f | fwd rel diff %.1e" % (
        c, n, res, timeit(f_own), timeit(f_mm), timeit(d_own), timeit(d_mm), timeit(w_own), timeit(w_mm), err))
