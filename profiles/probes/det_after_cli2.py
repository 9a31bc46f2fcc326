"""After the CLI test: D's gradients of step 0 (a penalty step) for three identically seeded Trainers."""
import argparse, os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), ROOT]
os.chdir(ROOT)
import torch
import test_hip_cli_gpu as t
if os.environ.get("SKIP_CLI") != "1":
    t.test_train_from_folder_on_gpu_bf16(pathlib.Path(tempfile.mkdtemp()))
sys.argv = ["bench.py"]
import bench, ops, hip_backend as hb
ops.set_precision("bf16")
torch.backends.cudnn.deterministic = True
a = argparse.Namespace(batch=16, image_size=128, gae=2, classifier="resnet", workdir="/tmp/sb_det", precision="bf16")
res = []
for k in range(4):
    bench.seed_all(42)
    tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
    tr.train()
    torch.cuda.synchronize()
    w0 = {n: float(p.detach().double().abs().sum()) for n, p in tr.StylEx.D.named_parameters()}
    res.append({n: (float(p.grad.double().sum()), float(p.grad.double().abs().sum())) for n, p in tr.StylEx.D.named_parameters() if p.grad is not None})
    print(k, tr.d_loss, tr.g_loss)
    wres = globals().setdefault("wres", [])
    wres.append(w0)
    for _ in range(int(os.environ.get("MORE_STEPS", "4"))):
        tr.train()
    torch.cuda.synchronize()
    del tr
    torch.cuda.empty_cache()
for n in res[1]:
    vals = [r[n] for r in res[1:]]
    if len(set(vals)) > 1:
        print("DIFF", n, vals)
for n in wres[1]:
    vals = [r[n] for r in wres[1:]]
    if len(set(vals)) > 1:
        print("WDIFF", n, vals)
print("done")
