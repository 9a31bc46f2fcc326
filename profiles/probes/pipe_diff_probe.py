"""Where does the pipelined conv kernel's output differ from the per-tile kernel's?  (debug aid: element positions of
the mismatches by op: fwd, dgrad+gate tensor, dgrad+gate bit mask)"""
import collections, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
sys.path[:0] = [os.path.join(PKG, "stylex"), PKG]
import torch, hip_backend as hb
P = hb.BF16_ACT; dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
B, c, n, res = 2, 128, 128, 64
mk = lambda *sh: torch.randn(*sh, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
x, dy, gate = mk(B, c, res, res), mk(B, n, res, res), mk(B, c, res, res)
w = torch.randn(n, c, 3, 3, device=dev, generator=g) / (9 * c) ** 0.5
bias = torch.randn(n, device=dev, generator=g)
bits = (gate.permute(0, 2, 3, 1).float() > 0).reshape(B, res, res, c // 8, 8).to(torch.int32)
gmask = (bits * (2 ** torch.arange(8, device=dev, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous()
ops = {"fwd": lambda: hb.conv2d_fwd(x, w, 1, 1, P, bias=bias, lrelu=True),
       "dgrad+gate": lambda: hb.conv2d_bwd_data(dy, w, (B, c, res, res), 1, 1, P, gate=gate),
       "dgrad+gmask": lambda: hb.conv2d_bwd_data(dy, w, (B, c, res, res), 1, 1, P, gate_mask=gmask)}
for name, fn in ops.items():
    outs = {}
    for arm in ("0", "1"):
        os.environ["STYLEX_CONV_PIPE"] = arm
        outs[arm] = fn().float().permute(0, 2, 3, 1).contiguous()
    torch.cuda.synchronize()
    d = (outs["0"] - outs["1"]).abs()
    idx = (d > 0).nonzero()
    print(name, "max diff", float(d.max()), "num diff", len(idx), "of", d.numel())
    if len(idx):
        b, y, xx, ch = idx[0].tolist()
        print("   first", idx[0].tolist(), "tile", outs["0"][b, y, xx, ch].item(), "pipe", outs["1"][b, y, xx, ch].item())
        print("   ch%32", collections.Counter((idx[:, 3] % 32).tolist()).most_common(10))
        print("   x%32", collections.Counter((idx[:, 2] % 32).tolist()).most_common(10))
        print("   y%16", collections.Counter((idx[:, 1] % 16).tolist()).most_common(6))
