#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/probe_first_conv.py > gpurun_out/r06_d_probe_first_conv.txt 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "image_gradient or lpips or frozen or resnet" 2>&1 | tail -5 > gpurun_out/r06_d_pytest_subset.txt
bash tools/ab_env.sh STYLEX_IMAGE_GRAD=0 STYLEX_DEVICE_RNG=1 > gpurun_out/r06_d_ab_imagegrad_devrng.txt 2>&1
