#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -2; done > gpurun_out/r06_ab_pytest_x3.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_ab_smoke.txt 2>&1
