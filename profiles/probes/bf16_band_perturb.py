"""Is the bf16 trajectory of steps_gae2_alt chaotic at call 2?  Unfused losses, with one loss scaled by (1 + eps)."""
import os, sys, tempfile, pathlib, json
os.environ["STYLEX_FUSED_LOSSES"] = os.environ.get("STYLEX_FUSED_LOSSES", "0")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")]
import ops
import stylex_train as st
from conftest import load_golden
from test_host_logic_cpu import make_trainer, run_steps
g = load_golden("steps_gae2_alt")
gold = np.asarray(g["scalars"])[:, :4]
out = {}
for which, eps in (("none", 0.0), ("hinge", 1e-6), ("kl", 1e-6), ("l1", 1e-6), ("hinge", -1e-6), ("gen", 1e-6)):
    orig = (st.hinge_loss, st.classifier_kl_loss, ops.l1_mean, st.gen_hinge_loss)
    if which == "hinge":
        st.hinge_loss = lambda r, f, o=orig[0]: o(r, f) * (1 + eps)
    if which == "kl":
        st.classifier_kl_loss = lambda r, f, o=orig[1]: o(r, f) * (1 + eps)
    if which == "l1":
        ops.l1_mean = lambda a, b, o=orig[2]: o(a, b) * (1 + eps)
    if which == "gen":
        st.gen_hinge_loss = lambda f, r, o=orig[3]: o(f, r) * (1 + eps)
    ops.set_precision("bf16")
    tr, n = make_trainer(g, pathlib.Path(tempfile.mkdtemp()), device=torch.device("cuda:0"))
    rows = np.asarray(run_steps(tr, n))[:, :4]
    st.hinge_loss, st.classifier_kl_loss, ops.l1_mean, st.gen_hinge_loss = orig
    rel = np.abs(rows - gold) / np.maximum(1.0, np.abs(gold))
    print("%-6s eps %+.0e  kl per call %s   max rel err per call %s" % (which, eps, np.round(rows[:, 3], 4), np.round(rel.max(axis=1), 4)), flush=True)
