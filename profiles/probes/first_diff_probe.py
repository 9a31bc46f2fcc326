"""Which kernel call is the first whose OUTPUT differs between the first and the second Trainer of one process?
Every public hip_backend function is wrapped: its tensor outputs are checksummed (this synchronises per call) and logged
in call order; two identically seeded Trainers must produce the same log.
usage (GPU box): [STYLEX_STREAMS=0] python tools/probes/first_diff_probe.py [steps] [image_size] [batch]"""
import argparse
import inspect
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
argv, sys.argv = sys.argv, ["bench.py"]
import bench  # noqa: E402

sys.argv = argv
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
torch.backends.cudnn.deterministic = True
LOG = []
SKIP = {"load_library", "is_cl", "act_dtype", "conv_shape", "empty_cl", "torgb_ok", "pack_cache_clear", "timing_enable",
        "timing_report", "timing_layers", "to_cl"}


def tensors(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in tensors(x)]
    return []


def wrap(name, fn):
    def inner(*a, **k):
        out = fn(*a, **k)
        ins = [tuple(t.shape) for t in tensors(a)][:3]
        cs = [(float(t.double().sum()), float(t.double().abs().sum())) for t in tensors(out)]
        LOG.append((name, ins, cs))
        return out

    return inner


for name, fn in list(vars(hb).items()):
    if inspect.isfunction(fn) and fn.__module__ == hb.__name__ and not name.startswith("_") and name not in SKIP:
        setattr(hb, name, wrap(name, fn))

a = argparse.Namespace(batch=batch, image_size=size, gae=2, classifier="resnet", workdir="/tmp/sb_fd", precision="bf16")
logs = []
for t in range(2):
    bench.seed_all(42)
    tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
    LOG.clear()
    for _ in range(steps):
        tr.train()
    torch.cuda.synchronize()
    logs.append(list(LOG))
    print("trainer", t, "calls", len(LOG), "losses", tr.d_loss, tr.g_loss)
    del tr
    torch.cuda.empty_cache()
l0, l1 = logs
print("same length:", len(l0) == len(l1))
bad = 0
for i, (x, y) in enumerate(zip(l0, l1)):
    if x != y:
        print("DIFF at call", i, "\n  A", x, "\n  B", y)
        for j in range(max(0, i - 4), i):
            print("   before:", j, l0[j][0], l0[j][1])
        bad += 1
        if bad >= 3:
            break
print("first-diff probe done; differing calls shown:", bad)
