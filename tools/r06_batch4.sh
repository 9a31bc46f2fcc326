#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -s -k "config2_bf16_tracks_fp32_call_by_call" 2>&1 | tail -80 > gpurun_out/r06_f_teacher_forced.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "lpips or image_gradient or (conv_triad and (case25 or case26))" 2>&1 | tail -40 > gpurun_out/r06_f_pytest_lpips.txt
bash tools/ab_env.sh STYLEX_LPIPS_BF16=0 > gpurun_out/r06_f_ab_lpips_bf16.txt 2>&1
