#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_q_pytest_gpu.txt
bash tools/ab_env.sh STYLEX_LPIPS_STEM=0 > gpurun_out/r06_q_ab_lpips_stem.txt 2>&1
python tools/probe_lpips.py 2>&1 | grep -v "amdgpu.ids" | head -30 > gpurun_out/r06_q_probe_lpips.txt
