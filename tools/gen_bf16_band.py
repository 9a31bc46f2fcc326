"""Measured perturbation sensitivity of the bf16 speed mode on the steps_gae2_alt fixture -> tests/golden/bf16_band_gae2_alt.npz
(round-4 VERDICT item 3c: the band of test_bf16_step_band_vs_reference_golden is derived from this, as assert_trajectory's
is from the reference's own spread for fp32).  Runs ON THE GPU BOX (the HIP path measures itself; nothing of the reference
is needed: the fixture's golden scalars are only used for the scale):

    python tools/gen_bf16_band.py

Realisations: the default bf16 path, then eight runs that each change rounding in the last place somewhere — one loss
term scaled by (1 +- 1e-6) (hinge, generator hinge, KL, L1), and the arithmetic-neutral switches STYLEX_FUSED_LOSSES=0 and
STYLEX_ADAM_PACK=0.  Stored: every realisation's scalars (d, g, rec, kl per call) and
spread[call][scalar] = max over realisations of |x_r - x_default| / max(1, |golden|)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, tempfile, pathlib, json
import numpy as np, torch
ROOT = %r
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import ops
import stylex_train as st
from conftest import load_golden
from test_host_logic_cpu import make_trainer, run_steps
which, eps = os.environ.get("BAND_WHICH", "none"), float(os.environ.get("BAND_EPS", "0"))
if which == "hinge":
    o0 = st.hinge_loss; st.hinge_loss = lambda r, f: o0(r, f) * (1 + eps)
if which == "kl":
    o1 = st.classifier_kl_loss; st.classifier_kl_loss = lambda r, f: o1(r, f) * (1 + eps)
if which == "l1":
    o2 = ops.l1_mean; ops.l1_mean = lambda a, b: o2(a, b) * (1 + eps)
if which == "gen":
    o3 = st.gen_hinge_loss; st.gen_hinge_loss = lambda f, r: o3(f, r) * (1 + eps)
g = load_golden("steps_gae2_alt")
ops.set_precision("bf16")
tr, n = make_trainer(g, pathlib.Path(tempfile.mkdtemp()), device=torch.device("cuda:0"))
rows = run_steps(tr, n)
print("ROWS " + json.dumps(np.asarray(rows)[:, :4].tolist()))
print("GOLD " + json.dumps(np.asarray(g["scalars"])[:, :4].tolist()))
''' % ROOT

CASES = [("default", {}),
         ("hinge+1e-6", {"BAND_WHICH": "hinge", "BAND_EPS": "1e-6", "STYLEX_FUSED_LOSSES": "0"}),
         ("hinge-1e-6", {"BAND_WHICH": "hinge", "BAND_EPS": "-1e-6", "STYLEX_FUSED_LOSSES": "0"}),
         ("gen+1e-6", {"BAND_WHICH": "gen", "BAND_EPS": "1e-6", "STYLEX_FUSED_LOSSES": "0"}),
         ("kl+1e-6", {"BAND_WHICH": "kl", "BAND_EPS": "1e-6", "STYLEX_FUSED_LOSSES": "0"}),
         ("kl-1e-6", {"BAND_WHICH": "kl", "BAND_EPS": "-1e-6", "STYLEX_FUSED_LOSSES": "0"}),
         ("l1+1e-6", {"BAND_WHICH": "l1", "BAND_EPS": "1e-6", "STYLEX_FUSED_LOSSES": "0"}),
         ("unfused losses", {"STYLEX_FUSED_LOSSES": "0"}),
         ("torch Adam + lazy packs", {"STYLEX_ADAM_PACK": "0"})]


def main():
    rows, names, gold = [], [], None
    for name, env in CASES:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        r = [l for l in out.stdout.splitlines() if l.startswith("ROWS ")]
        if not r:
            print("FAILED", name, out.stderr[-800:])
            continue
        rows.append(json.loads(r[0][5:]))
        names.append(name)
        gold = json.loads([l for l in out.stdout.splitlines() if l.startswith("GOLD ")][0][5:])
    rows, gold = np.asarray(rows), np.asarray(gold)
    assert names and names[0] == "default"
    scale = np.maximum(1.0, np.abs(gold))
    spread = (np.abs(rows - rows[0][None]) / scale[None]).max(axis=0)
    err = np.abs(rows - gold[None]) / scale[None]
    np.set_printoptions(precision=4, suppress=True, linewidth=160)
    print("realisations:", names)
    print("error of the default bf16 path vs the fp32 golden, per call x (d, g, rec, kl):\n", err[0])
    print("spread of the bf16 path against itself (max over %d realisations):\n" % (len(names) - 1), spread)
    out = os.path.join(ROOT, "tests", "golden", "bf16_band_gae2_alt.npz")
    np.savez_compressed(out, names=np.array(names), rows=rows, golden=gold, spread=spread)
    print("wrote", out)


if __name__ == "__main__":
    main()
