"""Spread of the bf16 speed mode's first calls on the steps_gae2_alt fixture across arithmetic-neutral switches (each
changes rounding in the last place somewhere): how wide must the band of test_bf16_step_band_vs_reference_golden be?
GPU box: python tools/probes/bf16_band_spread.py"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, tempfile, pathlib, json
import numpy as np, torch
ROOT = %r
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import ops
from conftest import load_golden
from test_host_logic_cpu import make_trainer, run_steps
g = load_golden("steps_gae2_alt")
ops.set_precision("bf16")
tr, n = make_trainer(g, pathlib.Path(tempfile.mkdtemp()), device=torch.device("cuda:0"))
rows = run_steps(tr, n)
print("ROWS " + json.dumps(np.asarray(rows)[:, :4].tolist()))
print("GOLD " + json.dumps(np.asarray(g["scalars"])[:, :4].tolist()))
''' % ROOT
res = {}
gold = None
for fl, rf, ap in itertools.product("01", "01", "01"):
    env = dict(os.environ, STYLEX_FUSED_LOSSES=fl, STYLEX_RES_FOLD=rf, STYLEX_ADAM_PACK=ap)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    rows = [l for l in out.stdout.splitlines() if l.startswith("ROWS ")]
    if not rows:
        print("failed", fl, rf, ap, out.stderr[-500:])
        continue
    res[(fl, rf, ap)] = json.loads(rows[0][5:])
    gold = json.loads([l for l in out.stdout.splitlines() if l.startswith("GOLD ")][0][5:])
import numpy as np
gold = np.asarray(gold)
print("golden\n", gold)
allr = np.asarray(list(res.values()))
for k, v in res.items():
    rel = np.abs(np.asarray(v) - gold) / np.maximum(1.0, np.abs(gold))
    print("losses=%s fold=%s adam_pack=%s  max rel err per call:" % k, np.round(rel.max(axis=1), 4))
rel = np.abs(allr - gold[None]) / np.maximum(1.0, np.abs(gold))[None]
print("worst over the 8 realisations, per call x scalar (d, g, rec, kl):\n", np.round(rel.max(axis=0), 4))
print("spread between realisations (max - min) / scale:\n", np.round((allr.max(axis=0) - allr.min(axis=0)) / np.maximum(1.0, np.abs(gold)), 4))
