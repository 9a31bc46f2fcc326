#!/bin/bash
# GPU box, round 6 batch 1: suite, 1-rank DDP A/B, config-2 band fixture, steady-state kernel table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r06_c_pytest_gpu.txt
bash tools/ddp_1rank_ab.sh gpurun_out/r06_c_torchrun_1rank.txt
timeout 900 python tools/gen_bf16_band_config2.py 20 > gpurun_out/r06_c_band_config2.txt 2>&1
cp tests/golden/bf16_band_config2.npz gpurun_out/ 2>/dev/null
bash tools/trace_bench.sh r06_c_steady --steps 12 --warmup 6 --roofline-steps 0 --fp32-steps 0
MS=$(python -c "import json,re;l=open('gpurun_out/trace_r06_c_steady_bench.txt').read();print(8*json.loads(l[l.index('{'):])['ms_per_step'])")
python tools/prof_window.py gpurun_out/trace_r06_c_steady.csv.gz $MS 8 150 > gpurun_out/r06_c_steady_state_kernels.txt 2>&1
