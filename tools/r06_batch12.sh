#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/trace_bench.sh r06_u_steady --steps 12 --warmup 6 --roofline-steps 0 --fp32-steps 0
MS=$(python -c "import json,re;l=open('gpurun_out/trace_r06_u_steady_bench.txt').read();print(8*json.loads(l[l.index('{'):])['ms_per_step'])")
python tools/prof_window.py gpurun_out/trace_r06_u_steady.csv.gz $MS 8 120 > gpurun_out/r06_u_steady_state_kernels.txt 2>&1
python tools/trace_busy.py gpurun_out/trace_r06_u_steady.csv.gz 4 8 > gpurun_out/r06_u_step_busy.txt 2>&1
python tools/op_sites.py --top 45 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > gpurun_out/r06_u_op_sites_by_time.txt
