"""Failing scenario (CLI test, then determinism_check at 128 px): where do two identically seeded Trainers first differ?
Records checksums inside the tangent-pass gradient penalty and of D's gradients right before D's optimiser step."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), ROOT]
os.chdir(ROOT)
import torch
import test_hip_cli_gpu as t
import determinism_check
t.test_train_from_folder_on_gpu_bf16(pathlib.Path(tempfile.mkdtemp()))
import gp_tangent, stylex_train as st
LOG = []
def cs(x):
    return None if x is None else (float(x.detach().double().sum()), float(x.detach().double().abs().sum()))
of, ob = gp_tangent._DRealPenalty.forward, gp_tangent._DRealPenalty.backward
def fwd(ctx, real, layout, *params):
    out = of(ctx, real, layout, *params)
    LOG.append(("fwd", cs(real), cs(out[0]), cs(out[1])))
    return out
def bwd(ctx, g_out, g_norm):
    gr = ob.__wrapped__(ctx, g_out, g_norm) if hasattr(ob, "__wrapped__") else ob(ctx, g_out, g_norm)
    LOG.append(("bwd", cs(g_out), cs(g_norm)) + tuple(cs(g) for g in gr[2:]))
    return gr
gp_tangent._DRealPenalty.forward = staticmethod(fwd)
gp_tangent._DRealPenalty.backward = staticmethod(torch.autograd.function.once_differentiable(bwd))
oo = st.Trainer._opt_step
def opt_step(self, opt):
    if opt is self.StylEx.D_opt:
        LOG.append(("Dgrads",) + tuple(cs(p.grad) for p in self.StylEx.D.parameters()))
    return oo(self, opt)
st.Trainer._opt_step = opt_step
oh = st.hinge_loss
def hinge(real, fake):
    LOG.append(("hinge fwd", cs(real), cs(fake), (float((fake < 1).sum()), float((real > -1).sum()))))
    if fake.requires_grad:
        fake.register_hook(lambda g: LOG.append(("hinge gfake", cs(g))) or None)
        real.register_hook(lambda g: LOG.append(("hinge greal", cs(g))) or None)
    return oh(real, fake)
st.hinge_loss = hinge
marks = []
if os.environ.get("HOOK_D") == "1":
    od = st.Trainer._d_compute
    def dcomp(self, *a, **k):
        if not getattr(self, "_hooked", False):
            self._hooked = True
            for i, (n, p) in enumerate(self.StylEx.D.named_parameters()):
                p.register_hook(lambda g, n=n: LOG.append(("contrib " + n, cs(g))) or None)
        return od(self, *a, **k)
    st.Trainer._d_compute = dcomp
sys.argv = ["bench.py"]
import bench
ob2 = bench.build_trainer
def bt(*a, **k):
    marks.append(len(LOG))
    return ob2(*a, **k)
bench.build_trainer = bt
runs = determinism_check.run(steps=2, image_size=128, batch=16, trainers=3)
print("g_loss step0:", [r[0][0][1] for r in runs])
marks.append(len(LOG))
segs = [LOG[marks[i]:marks[i + 1]] for i in range(len(marks) - 1)][1:]   # drop the warm-up trainer
for ti, seg in enumerate(segs):
    print("trainer", ti, [(r[0], r[1:]) for r in seg if r[0].startswith("hinge")][:8])
    print("   kinds", [r[0][:14] for r in seg][:6], len(seg))
for j, recs in enumerate(zip(*segs)):
    kinds = {r[0] for r in recs}
    for k in range(1, len(recs[0])):
        vals = [r[k] for r in recs]
        if len(set(vals)) > 1:
            print("first difference: record", j, kinds, "field", k, vals)
            break
    else:
        continue
    break
else:
    print("no difference in the recorded quantities")
