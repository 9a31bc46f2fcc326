"""GPU probe: whole-step HIP-graph replay vs the eager enqueue of the same Trainer.

  python tools/graph_probe.py [--size 64 --batch 4 --steps 14]

Two Trainers are built from identical seeds with capturable optimisers; one never captures (graph_warmup = inf),
the other replays graphs after its warm-up.  Prints the loss trajectory of both and the largest relative
difference, then times eager vs graph steps.  Exit code 1 when the trajectories disagree."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
for p in (os.path.join(PKG, "stylex"), PKG, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def build(args, warm):
    import stylex_train as st

    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    ring = [torch.rand(args.batch, 3, args.size, args.size, generator=gen).to(dev) for _ in range(8)]
    seed_all(42)
    tr = st.Trainer(name="gp", base_dir="/tmp/stylex_graph_probe", image_size=args.size, network_capacity=16, fmap_max=512,
                    batch_size=args.batch, gradient_accumulate_every=args.gae, lr=2e-4, ttur_mult=1.5, mixed_prob=0.9,
                    rec_scaling=1, kl_scaling=1, aug_prob=0., classifier_name="resnet", classifier_path=None,
                    evaluate_every=10 ** 9, save_every=10 ** 9, tensorboard_dir=None, device=dev, graphs=True,
                    graph_warmup=warm)
    tr.loader = st.cycle(ring)
    tr.dataset = list(range(10 ** 6))
    tr.save = lambda *a, **k: None
    tr.evaluate = lambda *a, **k: None
    tr.init_StylEx()
    seed_all(43)
    return tr


def run(tr, n):
    rows = []
    for _ in range(n):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                     np.nan if tr.last_gp_loss is None else tr.last_gp_loss])
    return np.array(rows, dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--gae", type=int, default=2)
    ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--time-steps", type=int, default=20)
    a = ap.parse_args()
    import ops

    ops.set_precision(a.precision)
    torch.backends.cudnn.deterministic = True  # frozen MIOpen nets: same algorithm in both runs
    eager = run(build(a, 10 ** 9), a.steps)
    tr = build(a, 4)
    graph = run(tr, a.steps)
    rel = np.abs(eager - graph) / np.maximum(1e-3, np.abs(eager))
    rel = np.where(np.isnan(eager) & np.isnan(graph), 0.0, rel)
    for i in range(a.steps):
        print("step %2d eager %s\n        graph %s" % (i, np.array2string(eager[i], precision=5), np.array2string(graph[i], precision=5)))
    print("captured graphs:", sorted(tr._graph_cache.keys()), " max rel diff %.3e" % np.nanmax(rel))
    ok = np.nanmax(rel) < (2e-2 if a.precision == "bf16" else 1e-3)
    # timing
    for name, warm in (("eager", 10 ** 9), ("graph", 4)):
        t = build(a, warm)
        for _ in range(10):
            t.train()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h0 = time.perf_counter()
        for _ in range(a.time_steps):
            t.train()
        host = time.perf_counter() - h0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s: %.2f ms/step wall, %.2f ms/step host enqueue" % (name, dt / a.time_steps * 1e3, host / a.time_steps * 1e3))
    print("graph_probe", "OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
