#!/bin/bash
# usage (GPU box): bash tools/round_end.sh [tag]  -> GPU suite, smoke, refreshed profiles + PMC traffic, steady-state kernel table under gpurun_out/<tag>_*
TAG=${1:-r02_g}
set -x
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.txt 2>&1
bash tools/refresh_profiles.sh ${TAG}
bash tools/trace_bench.sh ${TAG}_steady --steps 12 --warmup 6 --roofline-steps 0 --fp32-steps 0
MS=$(python -c "import json,re;l=open('gpurun_out/trace_${TAG}_steady_bench.txt').read();print(8*json.loads(l[l.index('{'):])['ms_per_step'])")
python tools/prof_window.py gpurun_out/trace_${TAG}_steady.csv.gz $MS 8 70 > gpurun_out/${TAG}_steady_state_kernels.txt 2>&1
ls -la gpurun_out | tail -30
