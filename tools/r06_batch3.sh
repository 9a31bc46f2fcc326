#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/probe_first_conv.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r06_e_probe_first_conv.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r06_e_pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_e_bench.json 2>gpurun_out/r06_e_bench.err
