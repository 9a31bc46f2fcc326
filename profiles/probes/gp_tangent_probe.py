"""Per-parameter comparison of the tangent-pass gradient penalty with the double backward (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex"), os.path.join(ROOT, "oracle")]
import torch
import hip_backend as hb, ops, stylex_train as st, gp_tangent
hb.load_library()
DEV = torch.device("cuda:0")
prec, size = sys.argv[1], int(sys.argv[2])
ops.set_precision(prec)
torch.manual_seed(21)
D = st.DiscriminatorE(size, network_capacity=16, fmap_max=512).to(DEV)
with torch.no_grad():
    for p in D.parameters():
        if p.dim() == 1:
            p.normal_(0, 0.1)
b = 4
real = torch.rand(b, 3, size, size, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
a = torch.tensor([1.0, 0.0, 1.0, 1.0], device=DEV) / b
wh, wg = float(sys.argv[3]), float(sys.argv[4])  # weights of the hinge-like and the penalty term
res = {}
os.environ["STYLEX_GP_TANGENT"] = "2"
for mode in ("double", "tangent"):
    D.zero_grad()
    if mode == "double":
        ops.set_fast(False)
        x = real.clone().requires_grad_()
        out = D(x)
        norms = st.gradient_norms(x, out)
    else:
        ops.set_fast(True)
        out, norms = gp_tangent.d_real_with_norms(D, real)
    ops.set_fast(False)
    (wh * (out.float() * a).sum() + wg * 10 * ((norms - 1) ** 2).mean()).backward()
    res[mode] = (out.detach().float(), norms.detach(), {n: p.grad.clone() for n, p in D.named_parameters()})
print("out", res["double"][0].tolist(), res["tangent"][0].tolist())
print("norms", res["double"][1].tolist(), res["tangent"][1].tolist())
for n in res["double"][2]:
    x, y = res["double"][2][n].double(), res["tangent"][2][n].double()
    print("%-28s max|a| %.3e  err %.3e  rel-l2 %.3e" % (n, x.abs().max(), (x - y).abs().max(), ((x - y).norm() / (x.norm() + 1e-30))))
