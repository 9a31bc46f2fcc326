"""Which gradients differ between two identical runs of ONE bench-size train step (multi-stream race locator).
    python tools/probes/race_locator.py [steps]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch

    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    argv, sys.argv = sys.argv, ["bench.py"]
    import bench
    sys.argv = argv
    sys.path[:0] = [os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
    import hip_backend as hb
    import ops

    hb.load_library()
    ops.set_precision("bf16")
    torch.backends.cudnn.deterministic = True
    a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir="/tmp/sb_loc", precision="bf16")
    runs = []
    for _ in range(2):
        bench.seed_all(42)
        tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
        for _i in range(steps):
            tr.train()
        torch.cuda.synchronize()
        g = {n: (p.grad.detach().double().sum().item(), p.grad.detach().double().abs().sum().item())
             for n, p in tr.StylEx.named_parameters() if p.grad is not None}
        runs.append(g)
        del tr
        torch.cuda.empty_cache()
    bad = [n for n in runs[0] if runs[0][n] != runs[1].get(n)]
    print("%d of %d gradients differ" % (len(bad), len(runs[0])))
    same = [n for n in runs[0] if n not in bad]
    print("identical:", " ".join(same))
    for n in bad:
        a_, b_ = runs[0][n][1], runs[1][n][1]
        print("  %-60s rel diff of |g| sum %.2e" % (n, abs(a_ - b_) / max(abs(a_), 1e-30)))


if __name__ == "__main__":
    os.chdir(ROOT)
    main()
