"""Stream-safety of the multi-stream train step: two fresh Trainers from the same seeds must produce bit-identical
losses and parameters (bench workload at 128 px, bf16 speed mode, HIP streams on, MIOpen pinned to deterministic
algorithms).  Every reduction of the HIP path has a fixed order, so any missing stream dependency would show up as
run-to-run noise.  tools/determinism_check.py is the full-size (256 px) version of the same check."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_two_runs_are_bit_identical_with_streams_on():
    import determinism_check

    assert os.environ.get("STYLEX_STREAMS", "1") != "0"
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        runs = determinism_check.run(steps=5, image_size=128, batch=16)  # step 0 and 4 carry the gradient penalty
    finally:
        os.chdir(cwd)
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert runs[0][1] == runs[1][1]


def test_two_runs_are_bit_identical_without_per_step_sync():
    """Same check with the host running ahead (no loss is read until the end): the configuration real training and
    bench.py run in, where a missing cross-stream dependency has the most room to show."""
    import determinism_check

    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        runs = determinism_check.run(steps=6, image_size=128, batch=16, lazy=True)
    finally:
        os.chdir(cwd)
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert runs[0][1] == runs[1][1]


def test_two_runs_are_bit_identical_at_the_benchmark_size():
    """The same check at the size bench.py runs (256 px, B=32, GAE=2): library GEMMs choose their algorithm by shape, and
    a split-K algorithm with atomic accumulation — run-to-run noise — only appeared at this size (K = B*H*W = 1M rows
    in the weight gradient of a 1x1 conv that ATen was briefly allowed to differentiate)."""
    import determinism_check

    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        runs = determinism_check.run(steps=3, image_size=256, batch=32)  # step 0: gradient penalty, 1-2: ordinary
    finally:
        os.chdir(cwd)
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert runs[0][1] == runs[1][1]
