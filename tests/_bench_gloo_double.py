"""Launcher used by tests/test_ddp_gloo.py::test_bench_py_two_ranks_over_gloo: bench.main() — the driver's multi-GPU
entry point — with the CPU test double installed as the op surface and gloo instead of RCCL (a fresh interpreter started
by torch.distributed.run has no GPU here).  Everything else is bench.py's own N > 1 code path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(2)
import bench  # noqa: E402
import ops  # noqa: E402
from cpu_ops import CpuOracleOps  # noqa: E402

ops.use_impl(CpuOracleOps)
bench.main(sys.argv[1:], backend="gloo", device=torch.device("cpu"))
