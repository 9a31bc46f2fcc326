import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import torch, torch.nn.functional as F
import hip_backend as hb, ops
hb.load_library(); ops.set_precision("bf16"); P = hb.BF16_ACT; DEV = "cuda:0"
B, H, W, C, N = 2, 128, 128, 64, 64
g = torch.Generator(device=DEV).manual_seed(35)
mk = lambda *sh: torch.randn(*sh, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
x, dy = mk(B, C, H, W), mk(B, N, H, W)
w = torch.randn(N, C, 3, 3, device=DEV, generator=g) / (9 * C) ** 0.5
bias = torch.randn(N, device=DEV, generator=g)
y = hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True)
os.environ["STYLEX_CONV_LINE64"] = "0"
y0 = hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True)
yr = F.leaky_relu(F.conv2d(x.float(), w.to(torch.bfloat16).float(), bias, 1, 1), 0.2)
for name, a in (("line64", y), ("pipe", y0)):
    d = (a.float() - yr).abs()
    rel = d / yr.abs().clamp_min(0.25)
    bad = rel > 2.0 ** -7
    print(name, "max abs", float(d.max()), "n bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()[:20]
        print(idx.tolist())
        bb = bad.nonzero()
        for dim, nm in enumerate("bcyx"):
            vals, cnt = torch.unique(bb[:, dim], return_counts=True)
            print(nm, list(zip(vals.tolist()[:40], cnt.tolist()[:40])))
d = (y.float() - y0.float()).abs()
print("line64 vs pipe: n differing", int((d > 0).sum()), "max", float(d.max()))
