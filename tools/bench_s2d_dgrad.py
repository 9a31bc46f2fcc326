"""Data gradient of the space-to-depth stride-2 conv (DiscriminatorBlock / encoder down-sampling): the round-5 kernel with
all four sub-positions per block (conv_s2d_dgrad.hip) against the per-sub-position halo kernel it replaces
(STYLEX_S2D_DGRAD=0) — results compared with each other and with the fp64 definition (autograd of the stride-2 conv),
times by hipEvents.
Usage (GPU box): python tools/bench_s2d_dgrad.py [--batch 64] [--iters 20] [--check-only]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch  # noqa: E402

import hip_backend as hb  # noqa: E402

# (name, C, res of the conv's INPUT): the down-sampling convs of D and the encoder at 256 px
SHAPES = [("64->64@256", 64, 256), ("128->128@128", 128, 128), ("256->256@64", 256, 64), ("512->512@32", 512, 32)]


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


def ref_dgrad_s2d(dy, w):
    """fp64: gradient of conv2d(x, w, stride 2, pad 1) w.r.t. x, stored space-to-depth: [b, (sy*2+sx)*C + c, y, x]"""
    b, n, h, _ = dy.shape
    c = w.shape[1]
    x = torch.zeros(b, c, 2 * h, 2 * h, dtype=torch.float64, device=dy.device, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w.double(), stride=2, padding=1)
    (gx,) = torch.autograd.grad(y, x, dy.double())
    return gx.view(b, c, h, 2, h, 2).permute(0, 3, 5, 1, 2, 4).reshape(b, 4 * c, h, h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--new-only", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="times only (ablation builds through STYLEX_HIP_LIB)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    prec, dev = hb.BF16_ACT, "cuda:0"
    torch.manual_seed(0)
    print("%-14s %5s | %9s %9s | %7s %7s | speed-up" % ("layer", "B", "new ms", "old ms", "new TF", "old TF"))
    for (name, c, res) in SHAPES:
        if a.only and a.only not in name:
            continue
        h = res // 2
        w = (torch.randn(c, c, 3, 3, device=dev) * 0.05).bfloat16().float()
        _, wb2 = hb.pack_weight_s2d(w)
        ws, xs = (c, 4 * c, 3, 3), lambda b: (b, 4 * c, h, h)
        for b in (() if a.no_check else (1, 3)):  # odd batch: the static tile list ends ragged
            dy = torch.randn(b, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            ref = ref_dgrad_s2d(dy.float(), w)
            outs = {}
            for mode in ("1", "0"):
                os.environ["STYLEX_S2D_DGRAD"] = mode
                outs[mode] = hb.conv2d_bwd_data(dy, None, xs(b), 1, 1, prec, packed=wb2, w_shape=ws, s2d_c=c).double()
                torch.cuda.synchronize()
            scale = ref.abs().max().item()
            e_new, e_old = ((outs["1"] - ref).abs().max() / scale).item(), ((outs["0"] - ref).abs().max() / scale).item()
            d_no = ((outs["1"] - outs["0"]).abs().max() / scale).item()
            print("%-14s B=%d | rel err vs fp64: new %.2e old %.2e | new vs old %.2e" % (name, b, e_new, e_old, d_no))
            assert e_new < 6e-3 and d_no < 6e-3, (e_new, e_old, d_no)  # bf16 output rounding: 2^-8 of the value
            assert d_no == 0.0 or h % 32 != 0, d_no  # >= 32-wide images: same K order as the old LDS-DMA kernel, bit-identical
        if a.check_only:
            continue
        for b in (a.batch // 2, a.batch, 2 * a.batch):
            dy = torch.randn(b, c, h, h, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            fn = lambda: hb.conv2d_bwd_data(dy, None, xs(b), 1, 1, prec, packed=wb2, w_shape=ws, s2d_c=c)
            t = {}
            for mode in (("1",) if a.new_only else ("1", "0", "1", "0")):
                os.environ["STYLEX_S2D_DGRAD"] = mode
                t[mode] = min(t.get(mode, 1e9), timeit(fn, a.iters))
            t.setdefault("0", float("nan"))
            fl = 2.0 * b * h * h * c * c * 9
            print("%-14s %5d | %9.3f %9.3f | %7.1f %7.1f | %.2fx" % (name, b, t["1"], t["0"], fl / t["1"] / 1e9, fl / t["0"] / 1e9,
                                                                    t["0"] / t["1"]))
    os.environ["STYLEX_S2D_DGRAD"] = "1"


if __name__ == "__main__":
    main()
