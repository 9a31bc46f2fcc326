#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_n_pytest_gpu.txt
bash tools/ab_env.sh STYLEX_RESIZE_FUSE=0 STYLEX_HALO_MIN_W=16 > gpurun_out/r06_n_ab_resize_halo.txt 2>&1
