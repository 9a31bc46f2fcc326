"""Forward of the space-to-depth stride-2 conv (DiscriminatorBlock / encoder down-sampling): the round-5 pipelined kernel
(conv_s2d_fwd.hip) against the kernels it replaces (STYLEX_S2D_FWD=0) — three forms: bias only, bias + residual tensor merge,
and the one-launch block tail with the 1x1 residual conv as a K segment — each compared with the fp64 definition and with
the old path, times by hipEvents.
Usage (GPU box): python tools/bench_s2d_fwd.py [--batch 64] [--iters 20] [--check-only]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import hip_backend as hb  # noqa: E402

# (name, C_res = block input channels, C = N = block output channels, res of the conv's INPUT)
SHAPES = [("64->64@256", 8, 64, 256), ("128->128@128", 64, 128, 128), ("256->256@64", 128, 256, 64), ("512->512@32", 256, 512, 32),
          ("512->512@16", 512, 512, 16)]


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


def s2d(x):
    b, c, h2, w2 = x.shape
    return x.view(b, c, h2 // 2, 2, w2 // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(b, 4 * c, h2 // 2, w2 // 2)


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--new-only", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    prec, dev = hb.BF16_ACT, "cuda:0"
    torch.manual_seed(0)
    print("%-14s %-8s %5s | %9s %9s | %7s %7s | speed-up" % ("layer", "form", "B", "new ms", "old ms", "new TF", "old TF"))
    for (name, cr, c, res) in SHAPES:
        if a.only and a.only not in name:
            continue
        h = res // 2
        w = (torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)).bfloat16().float()
        wres = (torch.randn(c, cr, device=dev) / cr ** 0.5).bfloat16()
        bias = torch.randn(c, device=dev)
        wf2, _ = hb.pack_weight_s2d(w)
        ws = (c, 4 * c, 3, 3)

        def forms(b, x, xs, r):
            x2 = cl(s2d(x).bfloat16())
            out = {"bias": lambda: hb.conv2d_fwd(x2, None, 1, 1, prec, bias=bias, packed=wf2, w_shape=ws, s2d_c=c),
                   "residual": lambda: hb.conv2d_fwd(x2, None, 1, 1, prec, bias=bias, residual=r, res_scale=0.7, packed=wf2, w_shape=ws,
                                                     s2d_c=c)}
            if cr % 8 == 0 and hb.s2d_res_supported(tuple(x2.shape), c, c, cr):
                out["merged"] = lambda: hb.conv2d_s2d_res_fwd(x2, wf2, xs, wres, bias, c, c, 0.7)
            return out

        for b in (1, 3):  # odd batch: the static tile list ends ragged
            x = torch.randn(b, c, res, res, device=dev).bfloat16().float()
            xs = cl(torch.randn(b, cr, h, h, device=dev).bfloat16())
            r = cl(torch.randn(b, c, h, h, device=dev).bfloat16())
            y0 = F.conv2d(x.double(), w.double(), bias.double(), stride=2, padding=1)
            refs = {"bias": y0, "residual": (y0 + r.double()) * 0.7,
                    "merged": (y0 + F.conv2d(xs.double(), wres.double()[:, :, None, None])) * 0.7}
            for mode in ("1", "0"):
                os.environ["STYLEX_S2D_FWD"] = mode
                hb._S2D_RES_OK.clear()
                for form, fn in forms(b, x, xs, r).items():
                    y = fn().double()
                    torch.cuda.synchronize()
                    err = ((y - refs[form]).abs().max() / refs[form].abs().max()).item()
                    print("%-14s %-8s B=%d | STYLEX_S2D_FWD=%s rel err vs fp64 %.2e" % (name, form, b, mode, err))
                    assert err < 8e-3, (name, form, mode, err)  # bf16 output rounding: 2^-8 of the value
        if a.check_only:
            continue
        for b in (a.batch // 2, a.batch, 2 * a.batch):
            x = torch.randn(b, c, res, res, device=dev).bfloat16().float()
            xs = cl(torch.randn(b, cr, h, h, device=dev).bfloat16())
            r = cl(torch.randn(b, c, h, h, device=dev).bfloat16())
            t = {}
            for mode in (("1",) if a.new_only else ("1", "0", "1", "0")):
                os.environ["STYLEX_S2D_FWD"] = mode
                hb._S2D_RES_OK.clear()
                for form, fn in forms(b, x, xs, r).items():
                    if form == "residual":
                        continue
                    t[(form, mode)] = min(t.get((form, mode), 1e9), timeit(fn, a.iters))
            for form in ("bias", "merged"):
                if (form, "1") not in t:
                    continue
                tn, to = t[(form, "1")], t.get((form, "0"), float("nan"))
                fl = 2.0 * b * h * h * c * (c * 9 + (cr if form == "merged" else 0))
                print("%-14s %-8s %5d | %9.3f %9.3f | %7.1f %7.1f | %.2fx" % (name, form, b, tn, to, fl / tn / 1e9, fl / to / 1e9, to / tn))
    os.environ["STYLEX_S2D_FWD"] = "1"


if __name__ == "__main__":
    main()
