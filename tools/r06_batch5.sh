#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "frozen or resnet or config2_bf16_tracks_fp32_call_by_call" 2>&1 | tail -30 > gpurun_out/r06_g_pytest_frozen.txt
bash tools/ab_env.sh STYLEX_FROZEN_BWD_BF16=0 > gpurun_out/r06_g_ab_frozen_bwd.txt 2>&1
bash tools/trace_bench.sh r06_g_steady --steps 12 --warmup 6 --roofline-steps 0 --fp32-steps 0
MS=$(python -c "import json,re;l=open('gpurun_out/trace_r06_g_steady_bench.txt').read();print(8*json.loads(l[l.index('{'):])['ms_per_step'])")
python tools/prof_window.py gpurun_out/trace_r06_g_steady.csv.gz $MS 8 110 > gpurun_out/r06_g_steady_state_kernels.txt 2>&1
