import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd")
for p in (ROOT, os.path.join(ROOT, "oracle"), PKG, os.path.join(PKG, "stylex")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "against_definition: compares kernels with their definition / the oracle; the kernel "
                                       "names it launches count as checked (see KERNELS_CHECKED below)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np

    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


# ---- kernel coverage of the GPU suite (round 6) -----------------------------------------------------------------------
# Every conv launcher names the kernel instantiation it launches (stylex_note_kernel, the spelling rocprofv3 prints).  Tests
# marked `against_definition` compare a kernel's result with the fp64 / fp32 definition, the CPU oracle, or bit for bit with
# a kernel that such a test covers; the names they launch are collected here, and tests/test_zz_kernel_coverage_gpu.py
# asserts that one benchmark step at BASELINE config 2 launches nothing outside that set — a selector threshold cannot hide
# an instantiation from the parity tests again (round-5 VERDICT, weak 1 / 3).
KERNELS_CHECKED = {}  # kernel name -> set of test ids that launched it
DEFINITION_TESTS_RUN = []


@pytest.fixture(autouse=True)
def _collect_checked_kernels(request):
    if request.node.get_closest_marker("against_definition") is None:
        yield
        return
    import hip_backend as hb

    hb.KERNELS_SEEN.clear()
    hb.timing_enable(1)
    try:
        yield
    finally:
        hb.timing_kernels()
        hb.timing_enable(0)
        for _, name in hb.KERNELS_SEEN:
            KERNELS_CHECKED.setdefault(name, set()).add(request.node.nodeid)
        DEFINITION_TESTS_RUN.append(request.node.nodeid)
