#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "image_gradient or layout_bridge or frozen_classifier or lpips or stem" 2>&1 | tail -5 > gpurun_out/r06_s_pytest_subset.txt
STYLEX_IMAGE_GRAD=2 python tools/probe_first_conv.py 2>&1 | grep -v "amdgpu.ids" | grep "image_grad\|==\|IMAGE_GRAD\|dilation\|Col2Im\|Cijk" > gpurun_out/r06_s_probe_first_conv.txt
bash tools/ab_env.sh STYLEX_IMAGE_GRAD=2 STYLEX_IMAGE_GRAD=0 > gpurun_out/r06_s_ab_imagegrad.txt 2>&1
