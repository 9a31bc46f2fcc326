"""TEST INFRASTRUCTURE — X1 learning-rate sweep (round-4 VERDICT item 3b).  Runs ONLY in the build container (imports
the reference through ``ref_shim``).

``curve_64_calm.npz`` pins 100 ``train()`` calls of the reference at lr = 1e-8, where parameters barely move: the test
sees schedule, RNG order and forward arithmetic but not a wrong gradient / optimiser step after the first calls.  This
script measures, for lr in {1e-7, 1e-6, 1e-5, 1e-4}, how far the REFERENCE stays from ITSELF (8 vs 4 intra-op threads:
only the CPU summation order changes) over the same 100-call window, and how far the parameters move — so that the
fixture can be pinned at the largest lr the reference holds to <= 1e-4 against itself (or the table shows that none
exists above 1e-8).

    python oracle/sweep_calm_lr.py [--lrs 1e-7,1e-6,1e-5,1e-4] [--n 100] [--out profiles/r05_x1_lr_sweep.txt]

Writes the table to --out and tests/golden/curve_64_lr_sweep.npz (scalars of both runs per lr, spread per call,
parameter movement)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import ref_shim  # noqa: E402


def run(st, lr, threads, n, start=4960, stop_at=None):
    size, cap, fmax, bs, gae = 64, 16, 512, 4, 2
    torch.set_num_threads(threads)
    cls = ref_shim.TinyClassifier(seed=99)
    gd = torch.Generator().manual_seed(7)
    batches = [torch.rand(bs, 3, size, size, generator=gd) for _ in range(8)]
    mg.seed_all(42)
    tr = ref_shim.make_reference_trainer(st, tempfile.mkdtemp(), cls, batches, image_size=size, network_capacity=cap,
                                         fmap_max=fmax, batch_size=bs, gradient_accumulate_every=gae, lr=lr,
                                         ttur_mult=1.5, rec_scaling=1, kl_scaling=1)
    tr.init_StylEx()
    tr.steps = start
    p0 = [p.detach().clone() for p in tr.StylEx.parameters()]
    rows, t0 = [], time.time()
    for i in range(n):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss,
                     tr.last_gp_loss if tr.last_gp_loss is not None else np.nan,
                     tr.pl_mean if tr.pl_mean is not None else np.nan])
        if i % 20 == 0:
            print("lr %.0e t%d call %d %s %.0fs" % (lr, threads, i, rows[-1][:2], time.time() - t0), flush=True)
        if stop_at is not None and i + 1 >= stop_at:
            break
    move = [float((p.detach() - q).abs().max()) for p, q in zip(tr.StylEx.parameters(), p0)]
    names, pst = mg.param_stats(tr.StylEx)
    return np.array(rows, dtype=np.float64), float(np.max(move)), float(np.median(move)), names, pst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lrs", default="1e-7,1e-6,1e-5,1e-4")
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "profiles", "r05_x1_lr_sweep.txt"))
    a = ap.parse_args()
    st = ref_shim.import_reference()
    keep = torch.get_num_threads()
    lines = ["X1 lr sweep: the reference Trainer.train() (64 px, capacity 16, B=4, GAE 2, steps 4960..) against ITSELF at 8 vs 4 "
             "intra-op threads; spread = max over the six scalars of |a-b| / max(|a|, 1e-3) per call",
             "%8s | %10s %10s %10s | %12s %12s | %s" % ("lr", "max spread", "call<=1e-4", "call<=1e-3", "max |dtheta|", "median", "calls")]
    out = {}
    for lr in [float(v) for v in a.lrs.split(",")]:
        ra, mv_max, mv_med, names, pst = run(st, lr, 8, a.n)
        rb, _, _, _, _ = run(st, lr, 4, a.n)
        m = min(len(ra), len(rb))
        rel = np.nanmax(np.abs(ra[:m] - rb[:m]) / np.maximum(np.abs(ra[:m]), 1e-3), axis=1)
        ok4 = int(np.argmax(rel > 1e-4)) if (rel > 1e-4).any() else m
        ok3 = int(np.argmax(rel > 1e-3)) if (rel > 1e-3).any() else m
        lines.append("%8.0e | %10.2e %10d %10d | %12.3e %12.3e | %d" % (lr, rel.max(), ok4, ok3, mv_max, mv_med, m))
        print(lines[-1], flush=True)
        key = "lr%.0e" % lr
        out[key + "/scalars_t8"], out[key + "/scalars_t4"], out[key + "/self_rel"] = ra, rb, rel
        out[key + "/move_max"], out[key + "/move_median"] = mv_max, mv_med
        out[key + "/param_stats"] = pst
        out["param_names"] = names
        open(a.out, "w").write("\n".join(lines) + "\n")
        mg.save("curve_64_lr_sweep", config=np.array([64, 16, 512, 4, 2, 1, a.n, 4960]), seed=42, data_seed=7, cls_seed=99,
                lpips_seed=4242, lrs=np.array([float(v) for v in a.lrs.split(",")]), **out)
    torch.set_num_threads(keep)


if __name__ == "__main__":
    main()
