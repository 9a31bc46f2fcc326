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


@pytest.mark.timeout(900)
def test_two_processes_sharing_the_gpu_stay_bit_identical():
    """Round 6: with a SECOND process on the same GPU identically seeded Trainers used to differ run to run (one in two to
    ten): the to-RGB data gradient lost a product in lanes 48-63 of some waves — a packed fp32 multiply with crossed operand
    halves that only the SLP-vectorised torgb.hip contained; never in isolation, never with the GPU to itself
    (profiles/r06_torgb_contention.txt; csrc/Makefile builds that file without the vectoriser now).  Two determinism
    checks at once, three Trainers each after the throw-away one, two train() calls at the benchmark size: the three
    parameter checksums of each process and its loss scalars agree."""
    import subprocess

    tool = os.path.join(ROOT, "tools", "determinism_check.py")
    env = dict(os.environ, DET_TRAINERS="3")
    procs = [subprocess.Popen([sys.executable, tool, "2", "256", "32"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=800))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    per_proc = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
        line = [ln for ln in so.splitlines() if ln.startswith("all parameter checksums")][-1]
        sums = [float(v) for v in line.split("[", 1)[1].rstrip("] \n").split(",")]
        assert "bit-identical: True" in so, so[-2000:]  # (the first two Trainers of the process, losses included)
        assert len(sums) == 3 and len(set(sums)) == 1, sums  # the defect: one Trainer in two to ten of a process differed
        per_proc.append(sums[0])
    # Across the two processes the values agreed in every run of the round as well; not asserted: on a fresh box MIOpen's
    # first-use solver choice for the frozen networks can differ between two processes that start together (seen at sizes a
    # box ran for the first time: both processes bit-stable, their values 2.5e-7 apart; equal from the second run on).
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_processes_one_gpu.txt"), "w") as f:
        f.write("parameter checksums of the two processes: %r (equal: %s)\n" % (per_proc, per_proc[0] == per_proc[1]))
