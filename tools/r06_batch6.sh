#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_h_pytest_gpu.txt
bash tools/ab_env.sh STYLEX_STREAMS=0 STYLEX_DBLOCK_SIDE=1 STYLEX_IMAGE_GRAD=0 STYLEX_DRAW_AHEAD=0 > gpurun_out/r06_h_ab_switches.txt 2>&1
