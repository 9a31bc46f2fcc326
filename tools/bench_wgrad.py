"""Weight-gradient kernels A/B: the round-5 pipelined LDS-DMA kernel (conv_wgrad_pipe.hip) against the kernels it
replaces (STYLEX_WGRAD_PIPE=0), per layer shape of the 256 px StylEx networks — results compared with each other and,
on small batches, with an fp64 torch reference; times by hipEvents.
Usage (GPU box): python tools/bench_wgrad.py [--batch 64] [--iters 20] [--check-only]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402

# (name, C, N, res) — the 3x3 / stride-1 layers of G, D and the encoder at 256 px with >= 64 channels on both sides
SHAPES = [
    ("64->64@256", 64, 64, 256),
    ("64->128@128", 64, 128, 128),
    ("128->128@128", 128, 128, 128),
    ("128->64@128", 128, 64, 128),
    ("64->64@128", 64, 64, 128),
    ("128->256@64", 128, 256, 64),
    ("256->256@64", 256, 256, 64),
    ("256->128@64", 256, 128, 64),
    ("128->128@64", 128, 128, 64),
    ("256->512@32", 256, 512, 32),
    ("512->512@32", 512, 512, 32),
    ("512->256@32", 512, 256, 32),
    ("256->256@32", 256, 256, 32),
    ("512->512@16", 512, 512, 16),
    ("64->32@256", 64, 32, 256),
    ("32->32@256", 32, 32, 256),
]


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


def ref_wgrad(x, dy, scale=None):
    """fp64 definition: dW[n][c][kh][kw] = sum dy[b,n,y,x] * (x*scale)[b,c,y+kh-1,x+kw-1]"""
    xd = x.double()
    if scale is not None:
        xd = xd * scale.double()[:, :, None, None]
    dyd = dy.double()
    xp = torch.nn.functional.pad(xd, (1, 1, 1, 1))
    b, c, h, w = xd.shape
    n = dyd.shape[1]
    out = torch.empty(n, c, 3, 3, dtype=torch.float64, device=x.device)
    for kh in range(3):
        for kw in range(3):
            out[:, :, kh, kw] = torch.einsum("bnyx,bcyx->nc", dyd, xp[:, :, kh:kh + h, kw:kw + w])
    return out


# (name, C, N, res) of the stride-2 convs whose input arrives space-to-depth: x2 [B, 4C, res/2, res/2]
S2D_SHAPES = [("s2 64->64@256", 64, 64, 256), ("s2 128->128@128", 128, 128, 128), ("s2 256->256@64", 256, 256, 64),
              ("s2 512->512@32", 512, 512, 32)]


def fold_ref(dw2, c):
    """stylex_fold_weight_grad_s2d in torch: dW[n][c][kh][kw] = dW2[n][(sy*2+sx)*C + c][kh2][kw2]"""
    n = dw2.shape[0]
    out = torch.empty(n, c, 3, 3, dtype=dw2.dtype, device=dw2.device)
    for kh in range(3):
        for kw in range(3):
            kh2, sy = (0, 1) if kh == 0 else (1, kh - 1)
            kw2, sx = (0, 1) if kw == 0 else (1, kw - 1)
            s_ = sy * 2 + sx
            out[:, :, kh, kw] = dw2[:, s_ * c:(s_ + 1) * c, kh2, kw2]
    return out


def s2d_section(a, prec, dev):
    print("---- space-to-depth stride-2 convs (folded weight gradient [N][C][3][3])")
    for (name, c, n, res) in S2D_SHAPES:
        if a.only and a.only not in name:
            continue
        h = res // 2
        if not a.no_check:
            b = 2
            x2 = torch.randn(b, 4 * c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            dy = torch.randn(b, n, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            ref = fold_ref(ref_wgrad(x2.float(), dy.float()), c)
            errs = {}
            for mode in ("1", "0"):
                os.environ["STYLEX_WGRAD_PIPE"] = mode
                hb._S2D_WGRAD_OK.clear()
                dw = hb.conv2d_bwd_weight_s2d(x2, dy, (n, c, 3, 3), prec)
                torch.cuda.synchronize()
                errs[mode] = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
            print("%-16s B=%d | rel err new %.2e old %.2e" % (name, b, errs["1"], errs["0"]))
            assert errs["1"] < 2e-5, errs
        if a.check_only:
            continue
        for b in (a.batch, 2 * a.batch):
            x2 = torch.randn(b, 4 * c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            dy = torch.randn(b, n, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            t = {}
            for mode in (("1",) if a.new_only else ("1", "0", "1", "0")):
                os.environ["STYLEX_WGRAD_PIPE"] = mode
                hb._S2D_WGRAD_OK.clear()
                fn = lambda: hb.conv2d_bwd_weight_s2d(x2, dy, (n, c, 3, 3), prec)
                t[mode] = min(t.get(mode, 1e9), timeit(fn, a.iters))
            t.setdefault("0", float("nan"))
            fl = 2.0 * b * h * h * n * c * 9
            print("%-16s %5d | %9.3f %9.3f | %7.1f %7.1f | %.2fx" % (name, b, t["1"], t["0"], fl / t["1"] / 1e9, fl / t["0"] / 1e9,
                                                                    t["0"] / t["1"]))
            del x2, dy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--new-only", action="store_true", help="time the new kernel only (PMC runs)")
    ap.add_argument("--only", default="")
    ap.add_argument("--skip-s2d", action="store_true")
    ap.add_argument("--np-ab", action="store_true", help="third arm: the new kernel with 64-channel tiles forced (STYLEX_WGRAD_PIPE_NP=1)")
    a = ap.parse_args()
    prec = hb.BF16_ACT
    dev = "cuda:0"
    torch.manual_seed(0)

    # ---- correctness first: small batches against the fp64 definition, with / without bias sums and an x scale
    worst = 0.0
    for (name, c, n, res) in ([] if a.no_check else SHAPES):
        if a.only and a.only not in name:
            continue
        for (b, with_scale, with_bias) in ((2, False, True), (3, True, False)):
            x = torch.randn(b, c, res, res, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            dy = torch.randn(b, n, res, res, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            sc = (torch.rand(b, c, device=dev) + 0.5) if with_scale else None
            ref = ref_wgrad(x.float(), dy.float(), sc)
            outs = {}
            for mode in ("1", "0"):
                os.environ["STYLEX_WGRAD_PIPE"] = mode
                r = hb.conv2d_bwd_weight(x, dy, (n, c, 3, 3), 1, 1, prec, x_scale=sc, want_bias_sum=with_bias)
                dw, db = r if with_bias else (r, None)
                torch.cuda.synchronize()
                err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
                outs[mode] = err
                if with_bias and db is not None:
                    dbr = dy.double().sum((0, 2, 3))
                    eb = ((db.double() - dbr).abs().max() / dbr.abs().max()).item()
                    outs[mode + "b"] = eb
            worst = max(worst, outs["1"], outs.get("1b", 0.0))
            print("%-14s B=%d scale=%d bias=%d | rel err new %.2e old %.2e%s" % (
                name, b, with_scale, with_bias, outs["1"], outs["0"],
                (" | bias new %.2e old %s" % (outs["1b"], ("%.2e" % outs["0b"]) if "0b" in outs else "n/a")) if "1b" in outs else ""))
    if not a.no_check:
        print("worst relative error of the new kernel: %.3e" % worst)
    assert worst < 2e-5, worst  # same bf16 operands, fp32 accumulation: only the summation order differs
    if not a.skip_s2d:
        s2d_section(a, prec, dev)
    os.environ["STYLEX_WGRAD_PIPE"] = "1"
    if a.check_only:
        return

    # ---- times
    print("%-14s %5s | %9s %9s | %7s %7s | speed-up" % ("layer", "B", "new ms", "old ms", "new TF", "old TF"))
    tot = [0.0, 0.0]
    for (name, c, n, res) in SHAPES:
        if a.only and a.only not in name:
            continue
        for b in ((a.batch, 2 * a.batch) if res >= 64 else (a.batch,)):
            if b * res * res * max(c, n) * 2 >= 1.4 * 2 ** 30:
                continue
            x = torch.randn(b, c, res, res, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            dy = torch.randn(b, n, res, res, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            t = {}
            for mode in (("1",) if a.new_only else ("1", "0", "1", "0")):
                os.environ["STYLEX_WGRAD_PIPE"] = mode
                fn = lambda: hb.conv2d_bwd_weight(x, dy, (n, c, 3, 3), 1, 1, prec, want_bias_sum=True)
                t[mode] = min(t.get(mode, 1e9), timeit(fn, a.iters))
            fl = 2.0 * b * res * res * n * c * 9
            t.setdefault("0", float("nan"))
            extra = ""
            if a.np_ab and n % 128 == 0:
                os.environ["STYLEX_WGRAD_PIPE"] = "1"
                os.environ["STYLEX_WGRAD_PIPE_NP"] = "1"
                t64 = min(timeit(fn, a.iters), timeit(fn, a.iters))
                os.environ.pop("STYLEX_WGRAD_PIPE_NP")
                tw_ = 32 if res >= 32 else 16
                stages = b * (res // tw_) * (res // (128 // tw_)) * (n // 128) * ((c + 63) // 64) / 256.0
                extra = " | 64-ch tiles %.3f ms (%.2fx of 128), stages/block %.1f" % (t64, t["1"] / t64, stages)
            print("%-14s %5d | %9.3f %9.3f | %7.1f %7.1f | %.2fx%s" % (name, b, t["1"], t["0"], fl / t["1"] / 1e9, fl / t["0"] / 1e9,
                                                                      t["0"] / t["1"], extra))
            if b == a.batch:
                tot[0] += t["1"]
                tot[1] += t["0"]
            del x, dy
    if tot[0] > 0:
        print("TOTAL at B=%d: new %.3f ms, old %.3f ms (%.2fx)" % (a.batch, tot[0], tot[1], tot[1] / tot[0]))


if __name__ == "__main__":
    main()
