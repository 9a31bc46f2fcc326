"""Does the step read memory it never wrote?  Fill (almost) all free device memory with a NaN bit pattern, give it back to
the driver, then run the bench workload: any never-written byte that is read now poisons a loss or a parameter.
usage (GPU box): [STYLEX_G_SIDE=1] python tools/probes/uninit_probe.py [steps] [image_size]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
argv, sys.argv = sys.argv, ["bench.py"]
import bench  # noqa: E402

sys.argv = argv
sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import hip_backend as hb  # noqa: E402
import ops  # noqa: E402

hb.load_library()
ops.set_precision("bf16")
torch.backends.cudnn.deterministic = True
dev = torch.device("cuda:0")
if os.environ.get("POISON", "1") == "1":
    free, total = torch.cuda.mem_get_info()
    chunks = []
    n = int(free * 0.9) // (1 << 30)
    for _ in range(n):
        chunks.append(torch.full(((1 << 30) // 4,), -1, dtype=torch.int32, device=dev))  # 0xFFFFFFFF: NaN as fp32 and bf16
    torch.cuda.synchronize()
    print("poisoned %d GiB" % n)
    del chunks
    torch.cuda.empty_cache()
a = argparse.Namespace(batch=32, image_size=size, gae=2, classifier="resnet", workdir="/tmp/sb_un", precision="bf16")
bench.seed_all(42)
tr = bench.build_trainer(a, dev, 0, 1)
for i in range(steps):
    tr.train()
    print(i, tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss, tr.last_gp_loss)
bad = [n for n, p in tr.StylEx.named_parameters() if not torch.isfinite(p).all()]
print("non-finite parameters:", len(bad), bad[:8])
print("checksum", float(sum(p.detach().double().abs().sum() for p in tr.StylEx.parameters())))
