"""Launcher used by tests/test_cli_cpu.py::test_cli_multi_gpus_through_torchrun_gloo: the product's cli.main() with the
CPU test double installed as the op surface (a fresh interpreter started by torch.distributed.run has no GPU here)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "explaining-in-style-reproducibility-study_amd", "stylex")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

torch.set_num_threads(2)
import cli  # noqa: E402
import ops  # noqa: E402
from cpu_ops import CpuOracleOps  # noqa: E402

ops.use_impl(CpuOracleOps)
cli.main()
