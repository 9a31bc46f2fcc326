"""Band of the full-size bf16-vs-fp32 trajectory test (tests/test_hip_parity.py::test_config2_bf16_trajectory_inside_the_fp32_self_spread;
round-5 VERDICT item 6) -> tests/golden/bf16_band_config2.npz.  Runs ON THE GPU BOX; nothing of the reference is needed —
the HIP fp32 parity mode (the mode every golden-vector test pins to the reference) measures its OWN sensitivity at BASELINE
config 2 (256 px, batch 32, GAE 2, ResNet-18, lr 2e-4):

    python tools/gen_bf16_band_config2.py [calls=20]

Realisations: the default fp32 run, then six runs that each start from ONE parameter element moved by one unit in the last
place (a weight of D, G, the encoder, the mapping network, G's constant input, D's last linear) — the smallest change a
different-but-equally-valid summation order could make.  Stored: every realisation's scalars per call (d, g, rec, kl, gp),
spread[call][scalar] = max over realisations of |x_r - x_0| / max(1, |x_0|), and the bf16 speed mode's rows from the same seeds
(a record: the test re-runs both modes itself)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
os.chdir(ROOT)

NAMES = ["default", "D.blocks[0].net[0].weight", "G.blocks[0].conv1.weight", "encoder.blocks[2].net[2].weight", "S first linear",
         "G.initial_block", "D.fc.weight"]


def pick(m, r):
    """The parameter realisation r nudges (r >= 1)."""
    S0 = next(p for p in m.S.parameters())
    return [None, m.D.blocks[0].net[0].weight, m.G.blocks[0].conv1.weight, m.encoder.blocks[2].net[2].weight, S0, m.G.initial_block,
            m.D.fc.weight][r]


def run(prec, calls, nudge=0, workdir="/tmp/band2"):
    import torch

    import bench
    import hip_backend as hb
    import ops

    ops.set_precision(prec)
    hb.pack_cache_clear()
    a = argparse.Namespace(batch=32, image_size=256, gae=2, classifier="resnet", workdir=workdir, precision=prec)
    bench.seed_all(42)
    tr = bench.build_trainer(a, torch.device("cuda:0"), 0, 1)
    if nudge:
        p = pick(tr.StylEx, nudge)
        with torch.no_grad():
            flat = p.view(-1)
            i = int(flat.abs().argmax())  # a normal number: (1 + 2^-23) moves it by exactly one ulp
            flat[i:i + 1].mul_(1.0 + 2.0 ** -23)
        hb.mark_updated([p])
    rows = []
    for _ in range(calls):
        tr.train()
        rows.append([tr.d_loss, tr.g_loss, tr.total_rec_loss, tr.total_kl_loss, tr.last_gp_loss])
    torch.cuda.synchronize()
    del tr
    torch.cuda.empty_cache()
    return np.array(rows, dtype=np.float64)


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    import hip_backend as hb

    hb.load_library()
    run("bf16", 2)  # the first Trainer of a process orders its double backward differently: warm the process
    fp32 = [run("fp32", calls, r) for r in range(len(NAMES))]
    bf16 = run("bf16", calls)
    fp32 = np.stack(fp32)
    scale = np.maximum(1.0, np.abs(fp32[0]))
    spread = (np.abs(fp32 - fp32[0][None]) / scale[None]).max(axis=0)
    np.set_printoptions(precision=4, suppress=True, linewidth=170)
    print("fp32 default rows (d, g, rec, kl, gp):\n", fp32[0])
    print("bf16 rows:\n", bf16)
    print("self-spread of the fp32 path over %d one-ulp realisations:\n" % (len(NAMES) - 1), spread)
    print("bf16 vs fp32, relative to max(1, |fp32|):\n", np.abs(bf16 - fp32[0]) / scale)
    out = os.path.join(ROOT, "tests", "golden", "bf16_band_config2.npz")
    np.savez_compressed(out, names=np.array(NAMES), rows_fp32=fp32, rows_bf16=bf16, spread=spread)
    print("wrote", out)


if __name__ == "__main__":
    main()
