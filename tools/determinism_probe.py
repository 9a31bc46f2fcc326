"""Where does run-to-run noise enter a step?  Two fresh Trainers, same seeds, one train() call each; compares
checksums of D's gradients, D's parameters after its step, the G-phase losses and G's gradients."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py"]
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb", precision="bf16")
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
if os.environ.get("STYLEX_DETERMINISTIC_LIBS", "1") == "1":
    torch.backends.cudnn.deterministic = True  # MIOpen (frozen classifier / LPIPS): reproducible algorithms only


def checks(named):
    return {n: float(t.detach().double().abs().sum()) for n, t in named if t is not None}


runs = []
for r in range(2):
    bench.seed_all(42)
    tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
    m = tr.StylEx
    rec = {}
    d_step, g_step = m.D_opt.step, m.G_opt.step

    def d_hook(*aa, **kk):
        rec["D.grad"] = checks((n, p.grad) for n, p in m.D.named_parameters())
        out = d_step(*aa, **kk)
        rec["D.param"] = checks(m.D.named_parameters())
        return out

    def g_hook(*aa, **kk):
        rec["G.grad"] = checks([(n, p.grad) for n, p in m.G.named_parameters()] +
                               [("enc." + n, p.grad) for n, p in m.encoder.named_parameters()])
        return g_step(*aa, **kk)

    m.D_opt.step, m.G_opt.step = d_hook, g_hook
    calls = {"G": [], "D": [], "E": [], "C": []}

    def wrap(mod, key):
        fwd = mod.forward

        def f(*aa, **kk):
            out = fwd(*aa, **kk)
            o = out[0] if isinstance(out, tuple) else out
            calls[key].append((tuple(float(t.detach().double().abs().sum()) for t in aa if torch.is_tensor(t)),
                               float(o.detach().double().abs().sum())))
            return out

        mod.forward = f

    wrap(m.G, "G"); wrap(m.D, "D"); wrap(m.encoder, "E")
    cls = tr.classifier.classify_images

    def cf(x):
        o = cls(x)
        calls["C"].append(((float(x.detach().double().abs().sum()),), float(o.detach().double().abs().sum())))
        return o

    tr.classifier.classify_images = cf
    rec["calls"] = calls
    tr.train()
    rec["loss"] = {"d": tr.d_loss, "g": tr.g_loss, "rec": tr.total_rec_loss, "kl": tr.total_kl_loss}
    runs.append(rec)
    del tr
    torch.cuda.empty_cache()
for k in ("E", "C", "G", "D"):
    for i, (x, y) in enumerate(zip(runs[0]["calls"][k], runs[1]["calls"][k])):
        print("call %s#%d inputs %s output %s" % (k, i, "same" if x[0] == y[0] else "DIFF %r %r" % (x[0], y[0]),
                                                    "same" if x[1] == y[1] else "DIFF %r %r" % (x[1], y[1])))
for key in ("D.grad", "D.param", "loss", "G.grad"):
    x, y = runs[0][key], runs[1][key]
    diff = [(n, x[n], y[n]) for n in x if x[n] != y[n]]
    print("%-8s %d/%d entries differ" % (key, len(diff), len(x)), diff[:4])
