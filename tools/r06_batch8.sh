#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/op_sites.py --by-count --top 90 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > gpurun_out/r06_j_op_sites_by_count.txt
python tools/op_sites.py --top 60 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > gpurun_out/r06_j_op_sites_by_time.txt
