#!/bin/bash
# usage (GPU box): bash tools/round_end.sh  -> GPU suite, smoke, refreshed profiles + PMC traffic, steady-state kernel table under gpurun_out/r02_f_*
set -x
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r02_f_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_f_smoke.txt 2>&1
bash tools/refresh_profiles.sh r02_f
bash tools/trace_bench.sh r02_f_steady --steps 12 --warmup 6 --roofline-steps 0
MS=$(python -c "import json,re;l=open('gpurun_out/trace_r02_f_steady_bench.txt').read();print(8*json.loads(l[l.index('{'):])['ms_per_step'])")
python tools/prof_window.py gpurun_out/trace_r02_f_steady.csv.gz $MS 8 70 > gpurun_out/r02_f_steady_state_kernels.txt 2>&1
ls -la gpurun_out | tail -30
